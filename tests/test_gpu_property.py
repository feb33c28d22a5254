"""Property-based GPU parity (hypothesis): random small indexes (tiny moduli -> long buckets and many
collisions, duplicated k-mers -> multi-node hits and frequency filtering), random ragged reads with N /
lower case, random k, filter threshold, reverse complements and code path — always bit-exact vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

ALPHABET = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)


@st.composite
def cases(draw):
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    k = draw(st.integers(1, 31))
    n_reads = draw(st.integers(1, 60))
    lens = rng.integers(0, draw(st.sampled_from([8, 40, 200, 1500])) + 1, size=n_reads)
    genome_len = int(lens.max()) + 64
    genome = rng.integers(0, 4, size=genome_len)
    offs = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    bases = np.empty(int(offs[-1]), dtype=np.uint8)
    for r in range(n_reads):
        s = int(rng.integers(0, genome_len - lens[r] + 1))
        seq = genome[s:s + lens[r]].copy()
        style = rng.integers(0, 3)
        letters = ALPHABET[seq + (4 if style == 1 else 0)]
        if style == 2 and lens[r]:
            letters[rng.integers(0, lens[r], size=max(1, lens[r] // 20))] = ord("N")
        bases[offs[r]:offs[r + 1]] = letters
    # index: k-mers drawn from the genome (so that reads hit), with duplicates under several nodes
    n_idx = draw(st.integers(1, 300))
    pos = rng.integers(0, genome_len - k + 1, size=n_idx)
    codes = genome.astype(np.uint64)
    kmers = np.zeros(n_idx, dtype=np.uint64)
    for j in range(k):
        kmers |= codes[pos + j] << np.uint64(2 * j)
    dup = rng.integers(0, n_idx, size=n_idx // 3)
    kmers = np.concatenate([kmers, kmers[dup]])
    n_nodes = draw(st.sampled_from([1, 7, 1000]))
    nodes = rng.integers(0, n_nodes, size=kmers.shape[0])
    modulo = draw(st.sampled_from([1, 2, 3, 17, 64, 257, 4099]))
    max_freq = draw(st.sampled_from([0, 1, 2, 3, 1000, 65535]))
    revcomp = draw(st.booleans())
    path = draw(st.sampled_from([1, 1, 2]))
    return dict(k=k, bases=bases, offs=offs, kmers=kmers, nodes=nodes, modulo=modulo, max_freq=max_freq,
                revcomp=revcomp, path=path, max_node=n_nodes - 1)


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
@given(cases())
def test_random_cases_bit_exact(oracle, c):
    from kmer_mapper_amd.engine import DeviceIndex, extract_kmers
    from kmer_mapper_amd.kmer_index import KmerIndex
    index = KmerIndex.from_flat_kmers(c["kmers"], c["nodes"], c["modulo"])
    expect, n = oracle.map_reads(index, c["max_node"], c["bases"], c["offs"], c["k"],
                                 max_index_lookup_frequency=c["max_freq"], also_revcomp=c["revcomp"])
    km = oracle.extract(c["bases"], c["offs"], c["k"])
    with DeviceIndex.from_index(index, c["max_node"]) as dev:
        dev.set_param("path", c["path"])
        if c["path"] == 2:
            try:
                dev.set_param("part_shift", 4)     # many partitions even for these tiny indexes ...
            except ValueError:
                pass                               # ... unless 16 buckets per slice do not fit this modulo
            assert dev.get_param("radix_available") == 1
        dev.map_reads(c["bases"], c["offs"], c["k"], c["max_freq"], also_revcomp=c["revcomp"])
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        dev.map_kmers(km, c["max_freq"], also_revcomp=c["revcomp"], k=c["k"])
        assert np.array_equal(dev.get_node_counts(), expect)
        assert np.array_equal(dev.in_index(km), oracle.in_index(index, km))
    assert np.array_equal(extract_kmers(c["bases"], c["offs"], c["k"]), km)


VALID = set(b"ACGTacgtNn")


def _records_model(raw, lpr=4):
    """What kmm_map_records must make of a FASTQ (lpr = 4 lines per record) or two-line FASTA (lpr = 2) chunk
    (include/kmm.h): complete lines in groups of lpr, '@' and '+' ('>') where they belong, every byte of a sequence line
    a nucleotide ('\\r' before the newline dropped).  Returns (consumed, reads) or (consumed, None) when the chunk must be
    refused."""
    lines, start = [], 0
    while True:
        e = raw.find(b"\n", start)
        if e < 0:
            break
        lines.append((start, e))
        start = e + 1
    n_rec = len(lines) // lpr
    consumed = lines[lpr * n_rec - 1][1] + 1 if n_rec else 0
    reads, ok = [], True
    for r in range(n_rec):
        (h0, _), (s0, s1) = lines[lpr * r], lines[lpr * r + 1]
        if raw[h0:h0 + 1] != (b"@" if lpr == 4 else b">"):
            ok = False
        if lpr == 4 and raw[lines[lpr * r + 2][0]:lines[lpr * r + 2][0] + 1] != b"+":
            ok = False
        seq = raw[s0:s1]
        if seq.endswith(b"\r"):
            seq = seq[:-1]
        if not set(seq) <= VALID:
            ok = False
        reads.append(seq)
    return consumed, (reads if ok else None)


@pytest.mark.parametrize("seed,lpr", [(11, 4), (12, 4), (13, 4), (14, 2), (15, 2)])
def test_damaged_fastq_is_refused_or_mapped_like_the_model(oracle, seed, lpr):
    """Fuzz of the GPU record parser (kmm_map_records, both paths): FASTQ and two-line FASTA chunks with bytes overwritten, newlines
    removed and inserted, lines dropped, the end cut anywhere.  Either the next synchronising call raises (the
    reference's reader raises on a malformed file: bnp.open(...).read_chunks, command_line_interface.py:102-111) or
    `consumed`, the record count and the node counts are exactly what the line model above gives — on the direct path
    (census + windows over raw bytes), on the radix path with the compaction into 2-bit flat reads done on the device, and
    with it done by the host threads (kmm_hostpack.hpp; a chunk they refuse goes to the device parser): never other counts."""
    from kmer_mapper_amd import _lib, synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rng = np.random.default_rng(seed)
    k = int(rng.choice([5, 11, 31]))
    index, genome = syn.make_index(4000, k=k, seed=500 + seed, plant=False)
    mx = index.max_node_id()
    g = syn.ACGT[genome]
    reads, pos = [], 0
    for i in range(1500):
        n = int(rng.choice([0, 3, k - 1, k, k + 1, 36, 100, 150, 151, 700]))
        reads.append(g[pos:pos + n].tobytes())
        pos = (pos + n + 3) % (len(g) - 1000)
    if lpr == 4:
        clean = b"".join(b"@r%d/1 len=%d\n" % (i, len(r)) + r + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"FI:@+#,5", dtype=np.uint8), size=len(r))) + b"\n"
                         for i, r in enumerate(reads))
    else:
        clean = b"".join(b">r%d len=%d\n" % (i, len(r)) + r + b"\n" for i, r in enumerate(reads))
    fmt = _lib.FORMAT_FASTQ if lpr == 4 else _lib.FORMAT_FASTA2
    pool = np.frombuffer(b"ACGTNacgtn@+>XZ-.*0 \t\n\n\n", dtype=np.uint8)
    n_refused = n_mapped = 0
    with DeviceIndex.from_index(index, mx) as dev:
        for trial in range(24):
            b = bytearray(clean)
            for _ in range(int(rng.integers(0, 4))):
                kind = int(rng.integers(0, 6))
                p = int(rng.integers(0, len(b) - 1))
                if kind == 0:                                   # one byte overwritten
                    b[p] = int(rng.choice(pool))
                elif kind == 1:                                 # a run overwritten
                    n = int(rng.integers(1, 300))
                    b[p:p + n] = bytes(rng.choice(pool, size=len(b[p:p + n])))
                elif kind == 2:                                 # the next newline removed
                    e = b.find(b"\n", p)
                    if e >= 0:
                        del b[e]
                elif kind == 3:                                 # a newline inserted
                    b[p:p] = b"\n"
                elif kind == 4:                                 # a whole line dropped
                    e0 = b.find(b"\n", p)
                    e1 = b.find(b"\n", e0 + 1) if e0 >= 0 else -1
                    if e1 >= 0:
                        del b[e0 + 1:e1 + 1]
                else:                                           # cut anywhere
                    del b[p:]
            raw_b = bytes(b)
            consumed, model = _records_model(raw_b, lpr)
            raw = np.frombuffer(raw_b, dtype=np.uint8) if raw_b else np.zeros(0, dtype=np.uint8)
            expect = None
            if model is not None:
                bases = np.frombuffer(b"".join(model), dtype=np.uint8)
                offs = np.concatenate([[0], np.cumsum([len(r) for r in model])]).astype(np.int64)
                expect, _ = oracle.map_reads(index, mx, bases, offs, k)
            for path, threads in ((1, 0), (2, 0), (2, 3)):      # direct kernel; device-side compaction; the host threads' packing
                dev.reset()
                dev.set_param("path", path)
                dev.set_param("host_pack_threads", threads)
                try:
                    used, n_rec = dev.map_records(raw, fmt=fmt, k=k) if raw.shape[0] else (0, 0)
                    got = dev.get_node_counts()
                except ValueError:
                    assert model is None, ("refused a chunk the line model accepts", trial, path)
                    n_refused += 1
                    dev.reset()
                    continue
                assert model is not None, ("mapped a chunk the line model refuses", trial, path)
                assert (used, n_rec) == (consumed, len(model)), (trial, path)
                assert np.array_equal(got, expect), (trial, path)
                n_mapped += 1
    assert n_refused and n_mapped
