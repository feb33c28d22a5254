"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the reference's known
answers and the committed goldens.  Bit-exact: this is integer work."""
import numpy as np
import pytest

from tests.helpers import batch, golden_small, index_from_vector, reference_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kmm():
    from kmer_mapper_amd import _lib
    assert _lib.device_count() >= 1, "GPU tests need a HIP device"
    import kmer_mapper_amd.engine as engine
    return engine


@pytest.fixture(scope="module")
def syn():
    from kmer_mapper_amd import synthetic
    return synthetic


# ---------------------------------------------------------------- reference known answers
@pytest.mark.parametrize("v", reference_vectors()["lookup"], ids=lambda v: v["name"])
def test_lookup_reference_vectors(kmm, v):
    from kmer_mapper_amd.mapper import in_graph_index, map_kmers_to_graph_index
    index = index_from_vector(v)
    q = np.array(v["query"], dtype=np.uint64)
    got = map_kmers_to_graph_index(index, v["max_node_id"], q, v["max_index_lookup_frequency"])
    assert got.dtype == np.uint32 and got.shape == (v["max_node_id"] + 1,)
    assert got.tolist() == v["expected_node_counts"]
    if "expected_in_index" in v:
        m = in_graph_index(index, q)
        assert m.dtype == np.uint8 and m.tolist() == v["expected_in_index"]


@pytest.mark.parametrize("v", reference_vectors()["string_lookup"], ids=lambda v: v["name"])
def test_reference_test_mapping_lowercase_query(kmm, v):
    """reference tests/test_mapping.py:33-40 through the drop-in entry points."""
    from kmer_mapper_amd.kmer_index import KmerIndex
    from kmer_mapper_amd.mapper import map_kmers_to_graph_index
    from kmer_mapper_amd.util import get_kmer_hashes_from_chunk_sequence
    node_kmers = get_kmer_hashes_from_chunk_sequence(batch(v["node_kmers"]), v["k"])
    index = KmerIndex.from_flat_kmers(node_kmers, np.arange(len(node_kmers), dtype=np.int64), v["modulo"])
    qk = get_kmer_hashes_from_chunk_sequence(batch([v["query"]]), v["k"])
    counts = map_kmers_to_graph_index(index, v["max_node_id"], qk, v["max_index_lookup_frequency"])
    assert counts.dtype == np.uint32 and counts.sum() == 1 and counts[v["expected_node"]] == 1
    counts = map_kmers_to_graph_index(index, v["max_node_id"], node_kmers, v["max_index_lookup_frequency"])
    assert counts[:len(node_kmers)].tolist() == v["expected_counts_of_node_kmers"]


def test_gpu_counter_reference_known_answer(kmm):
    """reference tests/test_gpucounter.py:41-48."""
    from kmer_mapper_amd.gpu_counter import GpuCounter
    kmers = np.array([1, 2, 3], dtype=np.uint64)
    nodes = np.array([10, 11, 12])
    counter = GpuCounter.from_kmers_and_nodes(kmers, nodes, 31)
    counter.initialize_cuda(2003)
    counter.count(np.array([1, 1, 1, 2, 3, 1, 3], dtype=np.uint64))
    node_counts = counter.get_node_counts(15)
    assert node_counts.dtype == np.float64 and len(node_counts) >= 15
    assert np.all(node_counts[[10, 11, 12]] == [4, 1, 2])


@pytest.mark.parametrize("v", reference_vectors()["extract"], ids=lambda v: v["name"])
def test_extract_known_answers(kmm, v):
    from kmer_mapper_amd.util import get_kmer_hashes_from_chunk_sequence
    got = get_kmer_hashes_from_chunk_sequence(batch(v["reads"]), v["k"])
    assert got.dtype == np.uint64 and got.tolist() == v["expected"]


# ---------------------------------------------------------------- committed goldens
@pytest.mark.parametrize("name", ["uniform", "ragged"])
def test_goldens(kmm, name):
    d, index, mx, k = golden_small()
    bases, offs = d[name + "_bases"], d[name + "_offsets"]
    km = kmm.extract_kmers(bases, offs, k)
    assert np.array_equal(km, d[name + "_kmers"])
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_kmers(km)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts"])
        dev.reset(); dev.map_reads(bases, offs, k)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts"])
        dev.reset(); dev.map_reads(bases, offs, k, max_index_lookup_frequency=2)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts_maxfreq2"])
        dev.reset(); dev.map_reads(bases, offs, k, max_index_lookup_frequency=65535)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts_nofilter"])
        dev.reset(); dev.map_reads(bases, offs, k, also_revcomp=True)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts_revcomp"])
        dev.set_param("path", 2); dev.set_param("part_shift", 5)      # radix-partitioned path
        dev.reset(); dev.map_reads(bases, offs, k)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts"])
        dev.reset(); dev.map_reads(bases, offs, k, also_revcomp=True)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts_revcomp"])
        dev.set_param("path", 1)
        dev.reset(); dev.map_kmers(km, also_revcomp=True, k=k)
        assert np.array_equal(dev.get_node_counts(), d[name + "_counts_revcomp"])
        assert np.array_equal(dev.in_index(km), d[name + "_in_index"])
        if name == "uniform":
            dev.reset(); dev.map_reads_uniform(bases, len(offs) - 1, 150, k)
            assert np.array_equal(dev.get_node_counts(), d[name + "_counts"])


# ---------------------------------------------------------------- seeded parity vs the oracle
@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("k", [1, 2, 5, 16, 31])
def test_fused_vs_oracle_ragged(kmm, syn, oracle, k, path):
    index, genome = syn.make_index(3000, k=k, seed=21 + k, plant=(k >= 16))
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 4000, 0, 220, seed=31 + k)
    expect, n = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", path)
        if path == 2:
            dev.set_param("part_shift", 4)      # hundreds of partitions even for this small index
            assert dev.get_param("n_partitions") > 64
        dev.map_reads(bases, offs, k)
        got = dev.get_node_counts()
    assert n > 0 and np.array_equal(got, expect)


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("read_len", [5, 16, 31, 32, 100, 150, 151, 4096, 5000])
def test_fused_vs_oracle_uniform(kmm, syn, oracle, read_len, path):
    k = min(31, read_len)
    index, genome = syn.make_index(5000, k=k, seed=41, plant=True)
    mx = index.max_node_id()
    n_reads = max(3, 300000 // read_len)
    bases, offs = syn.make_reads(genome, n_reads, read_len, seed=43)
    expect, n = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", path)
        if path == 2:
            dev.set_param("part_shift", 4)
        dev.map_reads_uniform(bases, n_reads, read_len, k)
        got_u = dev.get_node_counts()
        dev.reset()
        dev.map_reads(bases, offs, k)
        got_g = dev.get_node_counts()
    assert np.array_equal(got_u, expect) and np.array_equal(got_g, expect)


def test_operator_map_kmers_vs_oracle_and_accumulation(kmm, syn, oracle):
    index, genome = syn.make_index(20000, seed=51)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 3000, 150, seed=52)
    km = oracle.extract(bases, offs, 31)
    rng = np.random.default_rng(5)
    km = np.concatenate([km, rng.integers(0, 2 ** 62, size=100000, dtype=np.uint64)])
    expect = oracle.map_kmers(index, mx, km)
    from kmer_mapper_amd.mapper import map_kmers_to_graph_index
    assert np.array_equal(map_kmers_to_graph_index(index, mx, km), expect)
    # second call starts from zero again (mapper.pyx:37), handle cache reused
    assert np.array_equal(map_kmers_to_graph_index(index, mx, km), expect)
    # handle-level calls accumulate across chunks (command_line_interface.py:124-130)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for part in np.array_split(km, 7):
            dev.map_kmers(np.ascontiguousarray(part))
        assert np.array_equal(dev.get_node_counts(), expect)
        assert np.array_equal(dev.in_index(km), oracle.in_index(index, km))
    # the reference's per-chunk pattern through the façade, summed in HBM instead of on the host: one fetch at the end
    from kmer_mapper_amd.mapper import NodeCountAccumulator
    parts = [np.ascontiguousarray(p) for p in np.array_split(km, 5)]
    summed = sum(map_kmers_to_graph_index(index, mx, p).astype(np.uint64) for p in parts).astype(np.uint32)
    with NodeCountAccumulator(index, mx) as acc:
        for p in parts:
            assert map_kmers_to_graph_index(index, mx, p, accumulate_into=acc) is None
        assert np.array_equal(acc.node_counts(), summed) and np.array_equal(summed, expect)
    other, _ = syn.make_index(500, seed=53)
    with pytest.raises(ValueError):
        map_kmers_to_graph_index(other, other.max_node_id(), km, accumulate_into=acc)


def test_device_resident_inputs_and_bound_counts(kmm, syn, oracle):
    import torch
    index, genome = syn.make_index(10000, seed=61)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 5000, 150, seed=62)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    d_bases = torch.from_numpy(bases).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    counts = torch.zeros(mx + 1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.bind_counts(counts)
        dev.map_reads(d_bases, d_offs, 31)
        dev.map_reads_uniform(d_bases, 5000, 150, 31)
        dev.synchronize()
        got = counts.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, expect * 2)
        # unaligned device pointer: the kernel falls back to byte loads
        d_shift = torch.empty(d_bases.numel() + 3, dtype=torch.uint8, device="cuda")
        d_shift[3:] = d_bases
        torch.cuda.synchronize()
        counts.zero_(); torch.cuda.synchronize()
        dev.map_reads(d_shift[3:], d_offs, 31)
        dev.synchronize()
        assert np.array_equal(counts.cpu().numpy().view(np.uint32), expect)


def test_skewed_nodes_and_collisions(kmm, syn, oracle):
    """Heavy atomic contention (1000 nodes) and a tiny modulo (long buckets, many collisions)."""
    index, genome = syn.make_index(20000, seed=71, skewed=True, modulo=1009)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 3000, 150, seed=72)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_uint32_wraparound(kmm):
    """counts wrap modulo 2^32 like the reference's uint32 vector (mapper.pyx:37,68)."""
    import torch
    v = reference_vectors()["lookup"][0]
    index = index_from_vector(v)
    counts = torch.full((16,), -3, dtype=torch.int32, device="cuda")   # 0xFFFFFFFD
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, 15) as dev:
        dev.bind_counts(counts)
        dev.map_kmers(np.array(v["query"], dtype=np.uint64))
        dev.synchronize()
    got = counts.cpu().numpy().view(np.uint32)
    assert got[10] == 1 and got[11] == 0xFFFFFFFE and got[0] == 0xFFFFFFFD


# ---------------------------------------------------------------- edge cases and errors
def test_empty_and_degenerate_inputs(kmm, syn):
    index, genome = syn.make_index(100, k=5, seed=81, plant=False)
    with kmm.DeviceIndex.from_index(index) as dev:
        dev.map_kmers(np.zeros(0, dtype=np.uint64))
        dev.map_reads(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64), 5)
        dev.map_reads(np.zeros(0, dtype=np.uint8), np.zeros(4, dtype=np.int64), 5)   # 3 empty reads
        b = batch(["ACG", "T", ""])                                                   # all shorter than k
        dev.map_reads(b.bases, b.offsets, 5)
        assert dev.get_node_counts().sum() == 0
        assert dev.in_index(np.zeros(0, dtype=np.uint64)).shape == (0,)
    assert kmm.extract_kmers(np.zeros(0, np.uint8), np.zeros(1, np.int64), 5).shape == (0,)


def test_invalid_base_raises(kmm, syn):
    index, genome = syn.make_index(100, k=5, seed=82, plant=False)
    b = batch(["ACGTACGT", "ACGTXCGT"])
    with kmm.DeviceIndex.from_index(index) as dev:
        dev.map_reads(b.bases, b.offsets, 5)
        with pytest.raises(ValueError, match="offset 12"):
            dev.get_node_counts()
        dev.reset()
        ok = batch(["ACGTACGT"])
        dev.map_reads(ok.bases, ok.offsets, 5)
        dev.get_node_counts()                      # error state was cleared
    with pytest.raises(ValueError, match="offset 12"):
        kmm.extract_kmers(b.bases, b.offsets, 5)


def test_custom_lut(kmm, oracle, syn):
    """Legacy A,C,T,G = 0,1,2,3 order (reference kmer_mapper/encodings.py:26-28) via the LUT knob."""
    lut = np.full(256, 0xFF, dtype=np.uint8)
    for i, c in enumerate("ACTG"):
        lut[ord(c)] = lut[ord(c.lower())] = i
    index, genome = syn.make_index(500, k=7, seed=83, plant=False)
    bases, offs = syn.make_reads(genome, 200, 50, seed=84, n_rate=0.0)
    expect, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 7, lut=lut)
    with kmm.DeviceIndex.from_index(index) as dev:
        dev.map_reads(bases, offs, 7, lut=lut)
        assert np.array_equal(dev.get_node_counts(), expect)
    assert np.array_equal(kmm.extract_kmers(bases, offs, 7, lut=lut), oracle.extract(bases, offs, 7, lut=lut))


def test_argument_errors(kmm, syn):
    from kmer_mapper_amd.mapper import map_kmers_to_graph_index
    v = reference_vectors()["lookup"][1]
    index = index_from_vector(v)
    with pytest.raises(ValueError):                       # Cython: Buffer dtype mismatch
        map_kmers_to_graph_index(index, 15, np.array(v["query"], dtype=np.int64))
    with pytest.raises(ValueError):                       # node 14 > max_node_id 12
        kmm.DeviceIndex.from_index(index, 12)
    bad = index_from_vector(v)
    bad._n_kmers = bad._n_kmers.copy(); bad._n_kmers[1] = 100      # bucket reaches past the entries
    with pytest.raises(ValueError):
        kmm.DeviceIndex.from_index(bad, 15)
    with kmm.DeviceIndex.from_index(index, 15) as dev:
        with pytest.raises(ValueError):
            dev.map_reads(np.zeros(4, np.uint8), np.array([0, 4], np.int64), 32)
        with pytest.raises(ValueError):
            dev.map_reads(np.zeros(4, np.uint8), np.array([1, 4], np.int64), 3)


# ---------------------------------------------------------------- full-size properties
def test_config1_scale_and_linearity(kmm, syn, oracle):
    """BASELINE config 1 (10 k reads of 150 bp, k=31) bit-exact; and counts are additive over any
    split of the reads (the property the reference's additive reduce relies on)."""
    index, genome = syn.make_index(1000, seed=1)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 10000, 150, seed=2)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    assert n == 1200000
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads_uniform(bases, 10000, 150, 31)
        whole = dev.get_node_counts()
        dev.reset()
        for lo, hi in ((0, 1234), (1234, 7777), (7777, 10000)):
            dev.map_reads_uniform(bases[lo * 150:hi * 150], hi - lo, 150, 31)
        parts = dev.get_node_counts()
    assert np.array_equal(whole, expect) and np.array_equal(parts, expect)


def test_large_batch_properties(kmm, syn, oracle):
    """2 M reads (240 M windows) resident in HBM: checksum of counts equals the number of hits a
    sampled oracle run predicts exactly on the sample, and uniform == general path on the whole."""
    import torch
    index, genome = syn.make_index(200000, seed=91)
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    R = 2_000_000
    d_bases = syn.make_reads_torch(g_ascii, R, 150, seed=92)
    d_offs = torch.arange(R + 1, dtype=torch.int64, device="cuda") * 150
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 1)
        dev.map_reads_uniform(d_bases, R, 150, 31)
        a = dev.get_node_counts()
        dev.reset()
        dev.set_param("path", 2)
        dev.map_reads(d_bases, d_offs, 31)
        b = dev.get_node_counts()
        assert np.array_equal(a, b)
        dev.reset()
        dev.map_reads_uniform(d_bases, R, 150, 31, also_revcomp=True)
        c = dev.get_node_counts()
        dev.reset()
        dev.set_param("path", 1)
        dev.map_reads_uniform(d_bases, R, 150, 31, also_revcomp=True)
        assert np.array_equal(c, dev.get_node_counts())
        # first 20 k reads against the oracle
        sample = d_bases[:20000 * 150].cpu().numpy()
        expect, _ = oracle.map_reads(index, mx, sample, np.arange(20001, dtype=np.int64) * 150, 31,
                                     n_threads=4)
        dev.reset()
        dev.map_reads_uniform(d_bases[:20000 * 150], 20000, 150, 31)
        assert np.array_equal(dev.get_node_counts(), expect)
    rate = a.sum() / (R * 120)
    assert 0.12 < rate < 0.25


# ---------------------------------------------------------------- CLI end to end (row f-1)
@pytest.mark.parametrize("fmt,gz", [("fa", False), ("fq", True)])
def test_cli_map_end_to_end(kmm, syn, oracle, tmp_path, fmt, gz):
    """`kmer_mapper map -i idx.npz -f reads -o out`: output <out>.npy equals the oracle's counts."""
    from kmer_mapper_amd import reads_io
    from kmer_mapper_amd.command_line_interface import run_argument_parser
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(5000, seed=101)
    bases, offs = syn.make_ragged_reads(genome, 3000, 20, 200, seed=102)
    batch = ReadBatch(bases, offs)
    idx_path = str(tmp_path / "index.npz")
    index.to_file(idx_path)
    reads_path = str(tmp_path / ("reads." + fmt + (".gz" if gz else "")))
    (reads_io.write_fasta if fmt == "fa" else reads_io.write_fastq)(reads_path, batch, gz=gz)
    out = str(tmp_path / "node_counts")
    run_argument_parser(["map", "-i", idx_path, "-f", reads_path, "-o", out, "-k", "31", "-c", "20000"])
    got = np.load(out + ".npy")
    expect, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 31, n_threads=4)
    assert got.dtype == np.uint32 and np.array_equal(got, expect)
    # in-memory index object + output_file=None returns the array (command_line_interface.py:146-147)
    import argparse
    from kmer_mapper_amd.command_line_interface import map_bnp
    ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=reads_path, kmer_size=31,
                            n_threads=16, chunk_size=2500000, output_file=None, debug=None,
                            max_hits_per_kmer=1000, gpu=False, gpu_hash_map_size=0,
                            map_reverse_complements=True)
    got_rc = map_bnp(ns)
    expect_rc, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 31, also_revcomp=True)
    assert np.array_equal(got_rc, expect_rc)


@pytest.mark.parametrize("line_width,crlf,gz", [(60, False, False), (7, False, True), (80, True, False)])
def test_multi_line_fasta_is_unwrapped_on_the_gpu(kmm, syn, oracle, tmp_path, line_width, crlf, gz):
    """FASTA whose sequences are wrapped over several lines (what `bnp.open` reads too, reference
    command_line_interface.py:102,109): kmm_map_records(KMM_FORMAT_FASTA) unwraps the chunk into two-line FASTA on
    the GPU (k-mers span the line breaks, never the records) — chunk by chunk with the tail carried over, the last
    chunk flagged, CRLF line ends, through `kmer_mapper map` on plain and gzipped files, against the oracle and the
    host parser."""
    import argparse
    from kmer_mapper_amd import _lib, reads_io
    from kmer_mapper_amd.command_line_interface import map_bnp
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(6000, seed=141)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 2500, 1, 400, seed=142)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    path = str(tmp_path / ("reads.fa" + (".gz" if gz else "")))
    reads_io.write_fasta(path, ReadBatch(bases, offs), gz=gz, line_width=line_width)
    if crlf:
        data = open(path, "rb").read().replace(b"\n", b"\r\n")
        open(path, "wb").write(data)
    assert reads_io.sniff_format(path) == ("fasta", False)
    ns = argparse.Namespace(kmer_index=index, index_bundle=None, reads=path, kmer_size=31, n_threads=16, chunk_size=30000,
                            output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                            map_reverse_complements=False)
    assert np.array_equal(map_bnp(ns), expect)
    ns.host_parser = True
    assert np.array_equal(map_bnp(ns), expect)
    # the operator directly: one call with the last-chunk flag; without it the last record stays unconsumed
    import gzip
    raw = np.frombuffer((gzip.open if gz else open)(path, "rb").read(), dtype=np.uint8)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        used, n_rec = dev.map_records(raw, raw.shape[0], _lib.FORMAT_FASTA | _lib.FORMAT_LAST_CHUNK, 31)
        assert (used, n_rec) == (raw.shape[0], 2500)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        used, n_rec = dev.map_records(raw, raw.shape[0], _lib.FORMAT_FASTA, 31)
        assert n_rec == 2499 and bytes(raw[used:used + 1]) == b">" and used == reads_io.records_cut(raw, "fasta_ml")
        used2, n2 = dev.map_records(raw[used:], raw.shape[0] - used, _lib.FORMAT_FASTA | _lib.FORMAT_LAST_CHUNK, 31)
        assert (used2, n2) == (raw.shape[0] - used, 1)
        assert np.array_equal(dev.get_node_counts(), expect)


# ---------------------------------------------------------------- more edge cases
def _flat_index(syn, kmers, nodes, modulo):
    from kmer_mapper_amd.kmer_index import KmerIndex
    return KmerIndex.from_flat_kmers(np.asarray(kmers, dtype=np.uint64), np.asarray(nodes, dtype=np.int64), modulo)


def _sparse_flat_index(kmers, nodes, modulo):
    """The reference's index layout for a few k-mers in a huge table: entries grouped by kmer % modulo,
    hashes_to_index / n_kmers written only where a bucket is in use (np.zeros maps untouched pages lazily)."""
    import types
    kmers = np.asarray(kmers, dtype=np.uint64)
    nodes = np.asarray(nodes, dtype=np.int32)
    h = (kmers % np.uint64(modulo)).astype(np.int64)
    order = np.argsort(h, kind="stable")
    kmers, nodes, h = kmers[order], nodes[order], h[order]
    h2i = np.zeros(modulo, dtype=np.int32)
    nk = np.zeros(modulo, dtype=np.int32)
    uh, first, cnt = np.unique(h, return_index=True, return_counts=True)
    h2i[uh] = first.astype(np.int32)
    nk[uh] = cnt.astype(np.int32)
    _, inv, kc = np.unique(kmers, return_inverse=True, return_counts=True)
    index = types.SimpleNamespace(_hashes_to_index=h2i, _n_kmers=nk, _nodes=nodes, _kmers=kmers,
                                  _frequencies=np.minimum(kc[inv], 65535).astype(np.uint16), _modulo=int(modulo))
    index.max_node_id = lambda: int(nodes.max())
    return index


@pytest.mark.parametrize("modulo", [1, 2, 64, 1009, 2 ** 20, 2147483629])
def test_fastmod_exact_for_every_modulo_and_full_uint64_range(kmm, oracle, modulo):
    """kmers[i] % modulo (mapper.pyx:54) for k-mers over the whole uint64 range, odd moduli,
    powers of two, modulo 1 and a prime just below 2^31."""
    rng = np.random.default_rng(modulo)
    idx_kmers = rng.integers(0, 2 ** 64, size=300, dtype=np.uint64)
    idx_kmers[:5] = [0, 1, 2 ** 64 - 1, 2 ** 63, modulo]
    if modulo > 2 ** 22:
        # a prime just below 2^31 (the largest table the reference's int32 arrays allow, mapper.pyx:22-23): the two
        # modulo-sized tables are zero pages apart from the 300 buckets in use, built without touching the rest
        index = _sparse_flat_index(idx_kmers, np.arange(300), modulo)
    else:
        index = _flat_index(None, idx_kmers, np.arange(300), modulo)
    q = np.concatenate([idx_kmers, rng.integers(0, 2 ** 64, size=5000, dtype=np.uint64),
                        idx_kmers + np.uint64(modulo)])
    expect = oracle.map_kmers(index, 299, q)
    with kmm.DeviceIndex.from_index(index, 299) as dev:
        dev.map_kmers(q)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert np.array_equal(dev.in_index(q), oracle.in_index(index, q))
        if dev.get_param("radix_available"):      # the same division in pass 1 of the radix path
            dev.reset()
            dev.set_param("path", 2)
            dev.map_kmers(q)
            assert np.array_equal(dev.get_node_counts(), expect)


def test_large_modulo_table(kmm, oracle):
    """A 2^31-sized hash space is too big for CI; 50 M buckets still exercises 64-bit indexing."""
    modulo = 50_000_017
    rng = np.random.default_rng(9)
    idx_kmers = rng.integers(0, 2 ** 62, size=2000, dtype=np.uint64)
    index = _flat_index(None, idx_kmers, rng.integers(0, 77, size=2000), modulo)
    q = np.concatenate([idx_kmers, rng.integers(0, 2 ** 62, size=20000, dtype=np.uint64)])
    expect = oracle.map_kmers(index, 76, q)
    with kmm.DeviceIndex.from_index(index, 76) as dev:
        dev.map_kmers(q)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_long_buckets_frequency_filter_and_poly_a(kmm, syn, oracle):
    """One bucket holding every entry (modulo 1), a k-mer present 1500x (> 1000: filtered) and 700x
    (kept), queried by poly-A reads."""
    k = 31
    a_kmer = np.uint64(0)                                   # AAAA...A
    c_kmer = np.uint64(int("01" * 31, 2))                   # CCCC...C
    kmers = np.concatenate([np.full(1500, a_kmer), np.full(700, c_kmer),
                            np.arange(1, 200, dtype=np.uint64)])
    nodes = np.arange(kmers.shape[0]) % 50
    for modulo in (1, 7):
        index = _flat_index(None, kmers, nodes, modulo)
        reads = ["A" * 100, "C" * 64, "A" * 31 + "C" * 31, "N" * 40]
        b = batch(reads)
        for mf in (1000, 699, 65535):
            expect, _ = oracle.map_reads(index, 49, b.bases, b.offsets, k, max_index_lookup_frequency=mf)
            with kmm.DeviceIndex.from_index(index, 49) as dev:
                dev.map_reads(b.bases, b.offsets, k, max_index_lookup_frequency=mf)
                assert np.array_equal(dev.get_node_counts(), expect)
                if mf == 1000:
                    assert expect.sum() == 35 * 700      # only the 34 + 1 all-C windows (frequency 700) count


@pytest.mark.parametrize("path", [1, 2])
def test_many_tiny_reads_and_very_long_reads(kmm, syn, oracle, path):
    index, genome = syn.make_index(4000, k=3, seed=111, plant=False, modulo=67)
    mx = index.max_node_id()
    # thousands of reads of length 0..5: far more than 256 read starts inside one tile
    bases, offs = syn.make_ragged_reads(genome, 20000, 0, 5, seed=112)
    expect, n = oracle.map_reads(index, mx, bases, offs, 3)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", path)
        if path == 2:
            dev.set_param("part_shift", 4)
        dev.map_reads(bases, offs, 3)
        assert n > 0 and np.array_equal(dev.get_node_counts(), expect)
    # a few reads of 23-100 kbp (windows spanning many tiles, no boundary for whole tiles)
    index, genome = syn.make_index(30000, k=31, seed=113)
    mx = index.max_node_id()
    rng = np.random.default_rng(114)
    lens = np.array([100000, 1, 50000, 0, 31, 30, 23457], dtype=np.int64)
    offs = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    starts = rng.integers(0, genome.shape[0] - 100000, size=lens.shape[0])
    bases = np.concatenate([syn.ACGT[genome[s:s + l]] for s, l in zip(starts, lens)])
    expect, n = oracle.map_reads(index, mx, bases, offs, 31)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", path)
        dev.map_reads(bases, offs, 31)
        got = dev.get_node_counts()
    assert expect.sum() > 1000 and np.array_equal(got, expect)


@pytest.mark.parametrize("total_mod", [1, 15, 16, 17, 1023, 1024, 1025])
def test_tail_and_alignment(kmm, syn, oracle, total_mod):
    """Batches whose byte count is just below / at / above the 16-byte vector and 1024-position tile."""
    index, genome = syn.make_index(2000, k=5, seed=121, plant=False)
    mx = index.max_node_id()
    total = 3 * 1024 + total_mod
    bases = syn.ACGT[genome[:total]]
    offs = np.array([0, 7, 7, 1500, total], dtype=np.int64)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 5)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads(bases, offs, 5)
        assert np.array_equal(dev.get_node_counts(), expect)
        # one read of exactly k bases, and one of k-1
        dev.reset(); dev.map_reads(bases[:5], np.array([0, 5], dtype=np.int64), 5)
        e1, n1 = oracle.map_reads(index, mx, bases[:5], np.array([0, 5], dtype=np.int64), 5)
        assert n1 == 1 and np.array_equal(dev.get_node_counts(), e1)
        dev.reset(); dev.map_reads(bases[:4], np.array([0, 4], dtype=np.int64), 5)
        assert dev.get_node_counts().sum() == 0


# ---------------------------------------------------------------- GPU record parser (kmm_map_records)
def _fastq_bytes(batch_, crlf=False, tricky_quality=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(len(batch_)):
        seq = batch_.bases[batch_.offsets[i]:batch_.offsets[i + 1]].tobytes()
        qual = (b"@+>" * len(seq))[:len(seq)] if tricky_quality else b"I" * len(seq)
        out.append(b"@read" + str(i).encode() + b" len=" + str(len(seq)).encode() + eol + seq + eol + b"+" + eol + qual + eol)
    return np.frombuffer(b"".join(out), dtype=np.uint8)


def _fasta2_bytes(batch_):
    out = []
    for i in range(len(batch_)):
        seq = batch_.bases[batch_.offsets[i]:batch_.offsets[i + 1]].tobytes()
        out.append(b">r" + str(i).encode() + b" ACGTNXYZ\n" + seq + b"\n")
    return np.frombuffer(b"".join(out), dtype=np.uint8)


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("crlf", [False, True])
def test_map_records_fastq_whole_and_chunked(kmm, syn, oracle, crlf, path):
    from kmer_mapper_amd import _lib
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(4000, seed=131)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 3000, 0, 200, seed=132)
    rb = ReadBatch(bases, offs)
    raw = _fastq_bytes(rb, crlf=crlf)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", path)
        if path == 2:
            dev.set_param("part_shift", 5)
        used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTQ)
        assert used == raw.shape[0] and n_rec == 3000
        assert np.array_equal(dev.get_node_counts(), expect)
        # arbitrary cuts: the library reports where the last complete record ends
        dev.reset()
        pos, total_rec, step = 0, 0, 70001
        while pos < raw.shape[0]:
            end = min(pos + step, raw.shape[0])
            used, n_rec = dev.map_records(np.ascontiguousarray(raw[pos:end]), fmt=_lib.FORMAT_FASTQ)
            assert used > 0 and raw[pos + used - 1] == 10
            assert pos + used == raw.shape[0] or raw[pos + used] == ord("@")
            pos += used
            total_rec += n_rec
        assert total_rec == 3000
        assert np.array_equal(dev.get_node_counts(), expect)


def test_map_records_fasta2_and_errors(kmm, syn, oracle):
    from kmer_mapper_amd import _lib
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(2000, seed=141)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 1000, 0, 180, seed=142)
    raw = _fasta2_bytes(ReadBatch(bases, offs))
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTA2)
        assert (used, n_rec) == (raw.shape[0], 1000)
        assert np.array_equal(dev.get_node_counts(), expect)
        # incomplete record at the end is left to the caller
        dev.reset()
        used, n_rec = dev.map_records(np.ascontiguousarray(raw[:-5]), fmt=_lib.FORMAT_FASTA2)
        assert n_rec == 999 and raw[used - 1] == 10 and raw[used] == ord(">")
        dev.get_node_counts()
        # a chunk without a complete record consumes nothing
        used, n_rec = dev.map_records(np.frombuffer(b">r1\nACGT", dtype=np.uint8), fmt=_lib.FORMAT_FASTA2)
        assert (used, n_rec) == (0, 0)
        # multi-line FASTA is rejected, not silently mis-parsed
        dev.reset()
        dev.map_records(np.frombuffer(b">r1\nACGT\nACGT\n>r2\nAC\n", dtype=np.uint8), fmt=_lib.FORMAT_FASTA2)
        with pytest.raises(ValueError, match="record structure"):
            dev.get_node_counts()
        # a non-nucleotide in a SEQUENCE line is an error; in headers / quality it is not
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1 XYZ\nACGTACGTAC\n+\n!!!!XYZ!!!\n", dtype=np.uint8), k=5)
        dev.get_node_counts()
        dev.map_records(np.frombuffer(b"@r1\nACGTXCGTAC\n+\nIIIIIIIIII\n", dtype=np.uint8), k=5)
        with pytest.raises(ValueError, match="offset 8"):
            dev.get_node_counts()
        # FASTQ whose third line is not '+'
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1\nACGT\nACGT\nIIII\n", dtype=np.uint8), k=3)
        with pytest.raises(ValueError, match="record structure"):
            dev.get_node_counts()


def test_map_records_device_buffer_large(kmm, syn, oracle):
    """A 150 MB raw FASTQ chunk already in HBM (several super-tiles of the newline census)."""
    import torch
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(50000, seed=151)
    mx = index.max_node_id()
    n, L = 480000, 150
    bases, offs = syn.make_reads(genome, n, L, seed=152)
    rec = np.empty((n, 4 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:4] = np.frombuffer(b"@rd\n", dtype=np.uint8)
    rec[:, 4:4 + L] = bases.reshape(n, L)
    rec[:, 4 + L:4 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 4 + L + 3:4 + L + 3 + L] = ord("F")
    rec[:, -1] = 10
    raw = rec.reshape(-1)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    d_raw = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        used, n_rec = dev.map_records(d_raw, fmt=_lib.FORMAT_FASTQ)
        assert (used, n_rec) == (raw.shape[0], n)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_map_cpu_facade_returns_fresh_vector_per_chunk(kmm, syn, oracle):
    from kmer_mapper_amd.command_line_interface import map_cpu
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(3000, seed=161)
    bases, offs = syn.make_reads(genome, 500, 100, seed=162)
    expect, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 31)
    for _ in range(2):      # no accumulation between calls (mapper.pyx:37)
        got = map_cpu({"kmer_size": 31}, index, ReadBatch(bases, offs))
        assert got.dtype == np.uint32 and np.array_equal(got, expect)


def test_extract_operator_large_ragged_and_device_io(kmm, syn, oracle):
    import torch
    index, genome = syn.make_index(20000, seed=171)
    bases, offs = syn.make_ragged_reads(genome, 60000, 0, 260, seed=172)
    expect = oracle.extract(bases, offs, 31)
    got = kmm.extract_kmers(bases, offs, 31)
    assert np.array_equal(got, expect)
    d_out = torch.empty(expect.shape[0], dtype=torch.int64, device="cuda")
    d_b, d_o = torch.from_numpy(bases).cuda(), torch.from_numpy(offs).cuda()
    torch.cuda.synchronize()
    kmm.extract_kmers(d_b, d_o, 31, out=d_out)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), expect)
    with pytest.raises(ValueError):                 # wrong n_out never writes past the buffer
        kmm.extract_kmers(bases, offs, 31, out=np.empty(expect.shape[0] - 1, dtype=np.uint64))
    for kk in (1, 13):
        assert np.array_equal(kmm.extract_kmers(bases, offs, kk), oracle.extract(bases, offs, kk))


def test_stats_count_lookups_and_hits(kmm, syn, oracle):
    index, genome = syn.make_index(3000, seed=181)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 2000, 0, 200, seed=182)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads(bases, offs, 31)
        assert dev.get_stats() == (n, int(expect.sum()))
        dev.map_reads(bases, offs, 31, also_revcomp=True)
        e2, _ = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=True)
        assert dev.get_stats(reset=True) == (3 * n, int(expect.sum()) + int(e2.sum()))
        assert dev.get_stats() == (0, 0)
        km = oracle.extract(bases, offs, 31)
        dev.set_param("path", 2)
        dev.map_reads(bases, offs, 31)
        dev.map_kmers(km)
        assert dev.get_stats() == (2 * n, 2 * int(expect.sum()))


def test_beyond_4gib_batch_int64_positions(kmm, syn, oracle):
    """A single call over 4.8e9 read bytes (> 2^32): the 1.6e9-byte batch repeated three times must give
    exactly three times its counts on the uniform, general and partitioned (sub-batched) paths."""
    import torch
    index, genome = syn.make_index(100000, seed=191)
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    R, L = 10_700_000, 150                        # 1.605e9 bytes per copy, 4.815e9 in total
    one = syn.make_reads_torch(g_ascii, R, L, seed=192)
    big = one.repeat(3)
    assert big.numel() > 2 ** 32
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads_uniform(one, R, L, 31)
        base = dev.get_node_counts().astype(np.uint64)
        sample = one[:20000 * L].cpu().numpy()
        dev.reset(); dev.map_reads_uniform(one[:20000 * L], 20000, L, 31)
        expect, _ = oracle.map_reads(index, mx, sample, np.arange(20001, dtype=np.int64) * L, 31, n_threads=4)
        assert np.array_equal(dev.get_node_counts(), expect)
        for path, general in ((1, False), (1, True), (2, False)):
            dev.reset()
            dev.set_param("path", path)
            if general:
                offs = torch.arange(3 * R + 1, dtype=torch.int64, device="cuda") * L
                torch.cuda.synchronize()
                dev.map_reads(big, offs, 31)
            else:
                dev.map_reads_uniform(big, 3 * R, L, 31)
            got = dev.get_node_counts().astype(np.uint64)
            assert np.array_equal(got, 3 * base), (path, general)
        assert dev.get_stats()[0] >= 3 * 3 * R * 120


@pytest.mark.parametrize("n,modulo", [(0, 5), (1, 1), (1000, 7), (5000, 10007), (300000, 600011), (2000, 2 ** 20),
                                      (40000, 5_000_011), (3_000_000, 6_000_011)])   # 3-level scans
def test_gpu_index_builder_equals_oracle_builder(kmm, oracle, n, modulo):
    """kmm_build_index == oracle_build_index (FlatKmers -> from_flat_kmers(modulo) -> convert_to_int32, reference
    tests/test_mapping.py:36-38; stable sort by hash), array for array; the package's numpy builder agrees too."""
    from kmer_mapper_amd.kmer_index import KmerIndex
    rng = np.random.default_rng(n + modulo)
    kmers = rng.integers(0, 2 ** 62, size=n, dtype=np.uint64)
    if n > 10:
        kmers[rng.integers(0, n, size=n // 5)] = kmers[rng.integers(0, n, size=n // 5)]   # duplicates
        kmers[: min(n, 1500)] = kmers[0] if n == 5000 else kmers[: min(n, 1500)]           # one long run
    nodes = rng.integers(0, 2 ** 31 - 1, size=n)
    a = oracle.build_index(kmers, nodes, modulo)
    b = KmerIndex.from_flat_kmers_gpu(kmers, nodes, modulo)
    c = KmerIndex.from_flat_kmers(kmers, nodes, modulo)
    for attr in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_frequencies"):
        x, y, z = getattr(a, attr), getattr(b, attr), getattr(c, attr)
        assert x.dtype == y.dtype == z.dtype and np.array_equal(x, y) and np.array_equal(x, z), attr
    assert b._modulo == modulo


def test_wide_bucket_layout_parity(kmm, syn, oracle, monkeypatch):
    """Indexes beyond the bitmap threshold use 32-byte buckets with two inline entries; force that
    layout on small indexes (KMM_OCC_MAX_BYTES=0) and run the main scenarios against the oracle."""
    from kmer_mapper_amd import _lib
    from kmer_mapper_amd.kmer_index import KmerIndex
    from kmer_mapper_amd.util import ReadBatch
    monkeypatch.setenv("KMM_OCC_MAX_BYTES", "0")
    for modulo in (None, 257, 1):           # default load factor, long buckets, a single bucket
        index, genome = syn.make_index(3000, seed=201, modulo=modulo)
        mx = index.max_node_id()
        bases, offs = syn.make_ragged_reads(genome, 1500, 0, 220, seed=202)
        km = oracle.extract(bases, offs, 31)
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            assert dev.get_param("wide_buckets") == 1 and dev.get_param("occupancy_filter") == 0
            for mf, rc in ((1000, False), (2, False), (65535, True)):
                expect, _ = oracle.map_reads(index, mx, bases, offs, 31, max_index_lookup_frequency=mf,
                                             also_revcomp=rc)
                dev.reset(); dev.map_reads(bases, offs, 31, mf, also_revcomp=rc)
                assert np.array_equal(dev.get_node_counts(), expect)
                dev.reset(); dev.map_kmers(km, mf, also_revcomp=rc, k=31)
                assert np.array_equal(dev.get_node_counts(), expect)
            assert np.array_equal(dev.in_index(km), oracle.in_index(index, km))
            raw = _fastq_bytes(ReadBatch(bases, offs))
            dev.reset(); dev.map_records(raw, fmt=_lib.FORMAT_FASTQ)
            assert np.array_equal(dev.get_node_counts(), oracle.map_reads(index, mx, bases, offs, 31)[0])
    # the reference's known answers on this layout too
    for v in reference_vectors()["lookup"]:
        idx = index_from_vector(v)
        with kmm.DeviceIndex.from_index(idx, v["max_node_id"]) as dev:
            dev.map_kmers(np.array(v["query"], dtype=np.uint64), v["max_index_lookup_frequency"])
            assert dev.get_node_counts().tolist() == v["expected_node_counts"]


def test_golden_fastq_chunk_through_record_parser(kmm):
    from kmer_mapper_amd import _lib
    d, index, mx, k = golden_small()
    raw = d["ragged_fastq"]
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=k)
        assert used == raw.shape[0] and n_rec == len(d["ragged_offsets"]) - 1
        assert np.array_equal(dev.get_node_counts(), d["ragged_counts"])


def test_non_monotone_offsets_are_reported_at_sync(kmm, syn):
    index, genome = syn.make_index(200, k=5, seed=211, plant=False)
    bases = syn.ACGT[genome[:300]]
    with kmm.DeviceIndex.from_index(index) as dev:
        dev.map_reads(bases, np.array([0, 100, 50, 300], dtype=np.int64), 5)
        with pytest.raises(ValueError, match="not non-decreasing at read 1"):
            dev.get_node_counts()
        dev.reset()
        dev.map_reads(bases, np.array([0, 100, 200, 300], dtype=np.int64), 5)
        dev.get_node_counts()


def test_handles_do_not_leak_device_memory(kmm, syn):
    """Create / use / destroy many handles (all paths, staging growth): free HBM returns to its level."""
    import torch
    index, genome = syn.make_index(50000, seed=221)
    bases, offs = syn.make_ragged_reads(genome, 20000, 0, 200, seed=222)
    km = np.arange(100000, dtype=np.uint64)

    def cycle(n):
        for i in range(n):
            with kmm.DeviceIndex.from_index(index) as dev:
                dev.set_param("path", 1 + (i & 1))
                dev.map_reads(bases, offs, 31)
                dev.map_kmers(km)
                dev.in_index(km[:1000])
                dev.get_node_counts()
            kmm.extract_kmers(bases[:50000], np.array([0, 50000], dtype=np.int64), 31)

    cycle(3)                                  # warm up allocator pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(40)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "leaked %.1f MiB of HBM" % ((free0 - free1) / 2 ** 20)


def test_cli_tiny_chunk_size_grows_until_a_record_fits(kmm, syn, oracle, tmp_path):
    """`-c 64` is smaller than one FASTQ record: the raw chunker must grow instead of looping forever."""
    from kmer_mapper_amd import reads_io
    from kmer_mapper_amd.command_line_interface import run_argument_parser
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(2000, seed=231)
    bases, offs = syn.make_ragged_reads(genome, 300, 100, 400, seed=232)
    idx_path, fq, out = str(tmp_path / "i.npz"), str(tmp_path / "r.fq"), str(tmp_path / "o")
    index.to_file(idx_path)
    reads_io.write_fastq(fq, ReadBatch(bases, offs))
    run_argument_parser(["map", "-i", idx_path, "-f", fq, "-o", out, "-c", "64"])
    expect, _ = oracle.map_reads(index, index.max_node_id(), bases, offs, 31)
    assert np.array_equal(np.load(out + ".npy"), expect)


@pytest.mark.parametrize("env", [
    {"KMM_BLOOM_BYTES": "0", "KMM_OCC_SHIFT": "0"},        # per-bucket bitmap, 1 bit
    {"KMM_BLOOM_BYTES": "0", "KMM_OCC_SHIFT": "1"},        # fingerprint-keyed, 2 bits
    {"KMM_BLOOM_BYTES": "0", "KMM_OCC_SHIFT": "3"},        # 8 bits
    {"KMM_BLOOM_BYTES": "4"},                               # a one-word Bloom filter: saturated, passes everything
    {"KMM_BLOOM_BYTES": "256"},                             # heavily loaded Bloom filter
    {"KMM_BLOOM_BYTES": "1048576"},                         # sparse Bloom filter
    {"KMM_WIDE_BUCKETS": "1", "KMM_BLOOM_BYTES": "4096"},   # wide buckets behind a Bloom filter
    {"KMM_WIDE_BUCKETS": "1", "KMM_BLOOM_BYTES": "0", "KMM_OCC_SHIFT": "1"},   # wide buckets behind the bitmap
])
def test_prefilter_variants_never_drop_a_hit(kmm, syn, oracle, monkeypatch, env):
    """Whatever the filter flavour and its false-positive rate, results stay bit-exact: a filter may only
    reject k-mers that are not in the index."""
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    index, genome = syn.make_index(6000, seed=241)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 3000, 0, 200, seed=242)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    expect_rc, _ = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=True, max_index_lookup_frequency=2)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        assert dev.get_param("occupancy_filter") == 1
        assert dev.get_param("wide_buckets") == int(env.get("KMM_WIDE_BUCKETS", "0"))
        if "KMM_BLOOM_BYTES" in env and env["KMM_BLOOM_BYTES"] != "0":
            assert dev.get_param("bloom_filter_bytes") == int(env["KMM_BLOOM_BYTES"])
        else:
            assert dev.get_param("occupancy_bits_per_bucket") == 1 << int(env["KMM_OCC_SHIFT"])
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset(); dev.map_reads(bases, offs, 31, 2, also_revcomp=True)
        assert np.array_equal(dev.get_node_counts(), expect_rc)
        dev.reset(); dev.set_param("occupancy_filter", 0); dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_empty_index(kmm, oracle):
    """An index without entries: every lookup misses, on all layouts."""
    import types
    M = 1009
    index = types.SimpleNamespace(_hashes_to_index=np.zeros(M, np.int32), _n_kmers=np.zeros(M, np.int32),
                                  _nodes=np.zeros(0, np.int32), _kmers=np.zeros(0, np.uint64),
                                  _frequencies=np.zeros(0, np.uint16), _modulo=M)
    b = batch(["ACGTACGTACGTAACCGGTT", "TTTTTTTTTTTT"])
    km = np.arange(1000, dtype=np.uint64)
    with kmm.DeviceIndex.from_index(index, 5) as dev:
        dev.map_reads(b.bases, b.offsets, 5)
        dev.map_kmers(km)
        assert dev.get_node_counts().tolist() == [0] * 6
        assert dev.in_index(km).sum() == 0
        assert dev.get_stats() == (16 + 8 + 1000, 0)


def test_index_arrays_may_live_in_hbm(kmm, syn, oracle):
    """kmm_index_create accepts device pointers for the five index arrays (used in place, no staging)."""
    import torch
    index, genome = syn.make_index(5000, seed=251)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 2000, 100, seed=252)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    dev_arrays = [torch.from_numpy(a).cuda() for a in
                  (index._hashes_to_index, index._n_kmers)]
    km = torch.from_numpy(index._kmers.view(np.int64)).cuda()          # uint64 bit patterns
    nd = torch.from_numpy(index._nodes).cuda()
    fr = torch.from_numpy(index._frequencies.view(np.int16)).cuda().view(torch.uint16) \
        if hasattr(torch, "uint16") else None
    if fr is None:
        pytest.skip("torch without uint16")
    torch.cuda.synchronize()
    with kmm.DeviceIndex(dev_arrays[0], dev_arrays[1], index._modulo, km, nd, fr, mx) as dev:
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_gpu_index_builder_with_huge_buckets(kmm, oracle):
    """a k-mer with 100 000 hits (one bucket of 100 000+ entries) and a dense collision bucket must not cost
    O(bucket^2): buckets above 64 entries are ordered by bitonic sorts.  Result identical to the stable numpy
    construction (what from_flat_kmers does upstream, reference tests/test_mapping.py:36-38)."""
    from kmer_mapper_amd.kmer_index import KmerIndex
    rng = np.random.default_rng(77)
    modulo = 50_021
    base = rng.integers(0, 1 << 62, size=60_000, dtype=np.uint64)
    heavy = np.uint64(987654321987)
    coll = np.uint64(modulo) * rng.integers(1, 1 << 40, size=5000, dtype=np.uint64) + np.uint64(123)  # one bucket
    coll = np.concatenate([coll, coll[:700]])                                                     # with repeats
    kmers = np.concatenate([base, np.full(100_000, heavy, dtype=np.uint64), coll])
    perm = rng.permutation(len(kmers))
    kmers = kmers[perm]
    nodes = rng.integers(0, 1 << 20, size=len(kmers)).astype(np.int64)
    a = oracle.build_index(kmers, nodes, modulo)
    b = KmerIndex.from_flat_kmers_gpu(kmers, nodes, modulo)
    for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_frequencies"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a._frequencies.max() == 65535 and a._n_kmers.max() >= 100_000


def test_kmer_index_file_through_the_reference_read_side(kmm, oracle, golden_dir, tmp_path):
    """Row f-2: the committed Kmer Index .npz (upstream key set, int64 on disk [UPSTREAM-UNVERIFIED]) ->
    KmerIndex.from_file -> convert_to_int32 -> remove_ref_offsets (kmer_mapper/util.py:56-62) -> the HIP lookup, both
    paths, against the counts frozen in the file (oracle_map_kmers, mapper.pyx:53-69) and against the oracle itself;
    then `kmer_mapper map -i <file>` end to end on reads that contain the file's k-mers."""
    import os
    from kmer_mapper_amd.kmer_index import KmerIndex
    from kmer_mapper_amd.mapper import map_kmers_to_graph_index, in_graph_index, clear_cache
    path = os.path.join(golden_dir, "kmer_index_small.npz")
    d = np.load(path)
    ix = KmerIndex.from_file(path)
    with pytest.raises(ValueError):
        map_kmers_to_graph_index(ix, int(d["test_max_node_id"]), d["test_query_kmers"])   # int64 tables are refused
    ix.convert_to_int32()
    ix.remove_ref_offsets()
    mx, q = ix.max_node_id(), d["test_query_kmers"]
    assert np.array_equal(map_kmers_to_graph_index(ix, mx, q), d["test_expected_counts"])
    assert np.array_equal(map_kmers_to_graph_index(ix, mx, q, 65535), d["test_expected_counts_maxfreq_65535"])
    assert np.array_equal(in_graph_index(ix, q), d["test_expected_in_index"])
    clear_cache()
    with kmm.DeviceIndex.from_index(ix, mx) as dev:
        for p in (1, 2):
            dev.reset()
            dev.set_param("path", p)
            dev.map_kmers(q)
            assert np.array_equal(dev.get_node_counts(), d["test_expected_counts"]), p
    # the CLI on the file: reads spelled from the index's own k-mers (k = 31, first base in the lowest bits)
    k = 31
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for km in ix._kmers[::7][:100]:
        codes = (int(km) >> (2 * np.arange(k))) & 3
        reads.append(letters[codes].tobytes())
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * k + b"\n" for i, r in enumerate(reads)))
    from kmer_mapper_amd.command_line_interface import run_argument_parser
    out = tmp_path / "counts"
    run_argument_parser(["map", "-i", path, "-f", str(fq), "-k", str(k), "-o", str(out), "-g", "True"])
    got = np.load(str(out) + ".npy")
    b = np.frombuffer(b"".join(reads), dtype=np.uint8)
    expect, _ = oracle.map_reads(ix, mx, b, np.arange(len(reads) + 1, dtype=np.int64) * k, k)
    assert np.array_equal(got, expect) and expect.sum() > 0
