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
