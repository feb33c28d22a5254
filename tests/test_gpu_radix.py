"""GPU tests of the radix path (csrc/kmm_radix.hpp): bit-exact against the oracle and against the direct path,
per-k-mer counting mode (GpuCounter semantics, reference kmer_mapper/gpu_counter.py:23-37), slices whose
entries exceed the LDS capacity, sticky device-side errors."""
import numpy as np
import pytest

from tests.helpers import batch

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


@pytest.mark.parametrize("shift", [2, 4, 7, 12])
@pytest.mark.parametrize("revcomp", [False, True])
def test_radix_equals_oracle_at_every_slice_width(kmm, syn, oracle, shift, revcomp):
    """Same reads, slice widths from 4 to 4096 buckets: one coarse partition up to hundreds, items that span
    many blocks, fine partitions with and without entries."""
    index, genome = syn.make_index(20000, seed=301)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 30000, 0, 260, seed=302)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=revcomp, n_threads=4)
    km = oracle.extract(bases, offs, 31)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("part_shift", shift)
        dev.set_param("path", 2)
        assert dev.get_param("n_partitions") == -(-index._modulo // (1 << shift))
        dev.map_reads(bases, offs, 31, also_revcomp=revcomp)
        assert np.array_equal(dev.get_node_counts(), expect)
        lookups = (2 if revcomp else 1) * n
        # conservation through the passes: every k-mer pass 1 emits is gathered once by pass 2 and probed once by pass 3
        assert dev.get_param("radix_p2_kmers") == lookups
        assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == lookups   # dropped: empty-bucket filter
        assert dev.get_stats(reset=True) == (lookups, int(expect.sum()))
        dev.reset()
        dev.map_kmers(km, also_revcomp=revcomp, k=31)          # operator entry point through the same passes
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        dev.map_reads(bases, offs, 31, max_index_lookup_frequency=1, also_revcomp=revcomp)
        e1, _ = oracle.map_reads(index, mx, bases, offs, 31, max_index_lookup_frequency=1, also_revcomp=revcomp)
        assert np.array_equal(dev.get_node_counts(), e1)


def test_radix_counts_accumulate_across_calls_and_paths(kmm, syn, oracle):
    """Counts of radix calls (kept per entry until the next synchronising call) and of direct calls add up."""
    index, genome = syn.make_index(8000, seed=311)
    mx = index.max_node_id()
    parts = [syn.make_reads(genome, 5000, 150, seed=312 + i) for i in range(4)]
    expect = sum(oracle.map_reads(index, mx, b, o, 31)[0].astype(np.uint64) for b, o in parts)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for i, (b, o) in enumerate(parts):
            dev.set_param("path", 1 + (i & 1))
            dev.map_reads_uniform(b, 5000, 150, 31)
        assert np.array_equal(dev.get_node_counts().astype(np.uint64), expect)
        # a second round on top, reading the counts in between
        for i, (b, o) in enumerate(parts):
            dev.set_param("path", 2 - (i & 1))
            dev.map_reads(b, o, 31)
            dev.synchronize()
        assert np.array_equal(dev.get_node_counts().astype(np.uint64), 2 * expect)


def test_auto_path_picks_radix_for_large_batches_only(kmm, syn, oracle):
    index, genome = syn.make_index(50000, seed=321)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 40000, 150, seed=322)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        assert dev.get_param("radix_available") == 1
        dev.set_timing(True)
        dev.map_reads_uniform(bases[:1000 * 150], 1000, 150, 31)     # 150 k positions: below the threshold
        t = dev.get_timing()
        assert t["k_map_reads"][1] == 1 and t["k_rx_p1"][1] == 0
        dev.reset()
        dev.set_param("radix_min_units", 1_000_000)
        dev.map_reads_uniform(bases, 40000, 150, 31)                  # 6 M positions
        t = dev.get_timing()
        assert t["k_map_reads"][1] == 0 and t["k_rx_p1"][1] == 1 and t["k_rx_p3"][1] == 1
        assert np.array_equal(dev.get_node_counts(), expect)


def test_slice_with_more_entries_than_lds_capacity(kmm, oracle):
    """One bucket with 6000 entries (same k-mer under 6000 nodes) and a fine partition whose entries exceed
    the 4096 keys a slice keeps in LDS: those buckets are walked in HBM, results unchanged."""
    rng = np.random.default_rng(331)
    modulo = 16381
    base = rng.integers(0, 1 << 62, size=3000, dtype=np.uint64)
    heavy = np.uint64(123456789012345)
    dense = (np.uint64(modulo) * np.arange(1, 3001, dtype=np.uint64) + np.uint64(77))   # 3000 k-mers, one bucket
    kmers = np.concatenate([base, np.full(6000, heavy, dtype=np.uint64), dense])
    nodes = np.concatenate([np.arange(3000), rng.integers(0, 5000, size=6000), rng.integers(0, 5000, size=3000)])
    from kmer_mapper_amd.kmer_index import KmerIndex
    index = KmerIndex.from_flat_kmers(kmers, nodes.astype(np.int64), modulo)
    mx = int(nodes.max())
    q = np.concatenate([base[:500], np.full(7, heavy, dtype=np.uint64), dense[::3],
                        rng.integers(0, 1 << 62, size=2000, dtype=np.uint64)])
    for mf in (1000, 65535):
        expect = oracle.map_kmers(index, mx, q, mf)
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            dev.set_param("path", 2)
            dev.map_kmers(q, mf)
            assert np.array_equal(dev.get_node_counts(), expect), mf
            dev.reset()
            dev.set_param("part_shift", 3)
            dev.map_kmers(q, mf)
            assert np.array_equal(dev.get_node_counts(), expect), mf
            # one slice of 8192 buckets holds all 12 000 entries: the 8192-key variant of pass 3 (32-bit directory,
            # one workgroup per CU), the rest of the heavy bucket still walked in HBM
            dev.reset()
            dev.set_param("part_shift", 13)
            dev.map_kmers(q, mf)
            assert np.array_equal(dev.get_node_counts(), expect), mf
    assert oracle.map_kmers(index, mx, q, 65535).sum() >= 7 * 6000


@pytest.mark.parametrize("n_entries, keys_in_lds", [(3000, 4096), (21500, 4608), (30000, 8192)])
def test_slices_of_8192_buckets(kmm, syn, oracle, n_entries, keys_in_lds):
    """part_shift 13: a sparse table (every slice's entries fit 4096 keys: 16-bit LDS directory, two workgroups of
    pass 3 per CU), one at load factor 0.54 (4608 keys, the 1 B-k-mer index's shape) and a dense one (8192-key slices,
    32-bit directory, one workgroup per CU)."""
    index, genome = syn.make_index(n_entries, seed=341, modulo=40009)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 20000, 0, 260, seed=342)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=True, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("part_shift", 13)
        dev.set_param("path", 2)
        assert dev.get_param("radix_p3_keys_in_lds") == keys_in_lds
        dev.map_reads(bases, offs, 31, also_revcomp=True)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == 2 * n


def test_per_kmer_counting_mode(kmm, syn, oracle):
    """GpuCounter semantics (reference gpu_counter.py:23-37): a count per index k-mer, node counts = their
    segmented sum over `nodes` (np.bincount in the reference)."""
    from kmer_mapper_amd.gpu_counter import GpuCounter
    index, genome = syn.make_index(6000, seed=341, plant=True)
    bases, offs = syn.make_reads(genome, 8000, 150, seed=342)
    q = oracle.extract(bases, offs, 31)
    kmers, nodes = index._kmers.copy(), index._nodes.astype(np.int64)
    perm = np.random.default_rng(343).permutation(len(kmers))           # any entry order must work
    kmers, nodes = kmers[perm], nodes[perm]
    counter = GpuCounter.from_kmers_and_nodes(kmers, nodes, 31)
    counter.initialize_cuda(12011)
    half = len(q) // 2
    counter.count(q[:half])
    counter.count(q[half:])
    uq, uc = np.unique(q, return_counts=True)
    pos = np.searchsorted(uq, kmers)
    pos[pos >= len(uq)] = 0
    expect_k = np.where(uq[pos] == kmers, uc[pos], 0).astype(np.uint32)
    got_k = counter.get_kmer_counts()
    assert got_k.dtype == np.uint32 and np.array_equal(got_k, expect_k)
    node_counts = counter.get_node_counts()
    assert node_counts.dtype == np.float64
    assert np.array_equal(node_counts, np.bincount(nodes, expect_k, minlength=len(node_counts)))
    # with reverse complements (`-r`, command_line_interface.py:74): additive
    counter.count(q[:1000], count_revcomps=True)
    rc = oracle.revcomp(q[:1000], 31)
    uq2, uc2 = np.unique(np.concatenate([q, q[:1000], rc]), return_counts=True)
    pos = np.searchsorted(uq2, kmers)
    pos[pos >= len(uq2)] = 0
    assert np.array_equal(counter.get_kmer_counts(), np.where(uq2[pos] == kmers, uc2[pos], 0).astype(np.uint32))


def test_device_errors_are_sticky_until_reset(kmm, syn):
    """a chunk with an invalid base has already been counted when the error is reported, so the error
    stays on the handle until reset() clears counts and error together."""
    index, genome = syn.make_index(100, k=5, seed=351, plant=False)
    bad = batch(["ACGTACGT", "ACGTXCGT"])
    ok = batch(["ACGTACGT"])
    for path in (1, 2):
        with kmm.DeviceIndex.from_index(index) as dev:
            dev.set_param("path", path)
            dev.map_reads(bad.bases, bad.offsets, 5)
            with pytest.raises(ValueError, match="offset 12"):
                dev.synchronize()
            dev.map_reads(ok.bases, ok.offsets, 5)
            with pytest.raises(ValueError, match="offset 12"):       # still there: counts are polluted
                dev.get_node_counts()
            dev.reset()
            dev.map_reads(ok.bases, ok.offsets, 5)
            dev.get_node_counts()


def test_radix_records_mode_and_long_reads(kmm, syn, oracle):
    """Raw FASTQ bytes through pass 1 (records front end), including a chunk whose tile count is a multiple of
    the block size and one that is not."""
    from kmer_mapper_amd import _lib
    from kmer_mapper_amd.util import ReadBatch
    index, genome = syn.make_index(4000, seed=361)
    mx = index.max_node_id()
    for n_reads in (50, 3000, 20011):
        bases, offs = syn.make_ragged_reads(genome, n_reads, 0, 200, seed=362 + n_reads)
        expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
        lines = []
        for r in range(n_reads):
            s = bases[offs[r]:offs[r + 1]].tobytes()
            lines.append(b"@r%d\n" % r + s + b"\n+\n" + b"I" * len(s) + b"\n")
        raw = np.frombuffer(b"".join(lines), dtype=np.uint8)
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            dev.set_param("path", 2)
            for threads in (0, 3):          # the device-side compaction; the host threads' packing (kmm_hostpack.hpp)
                dev.set_param("host_pack_threads", threads)
                dev.reset()
                used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTQ)
                assert used == raw.shape[0] and n_rec == n_reads
                assert np.array_equal(dev.get_node_counts(), expect), threads
            assert dev.get_param("host_packed_record_calls") == 1


def _fastq(reads, eol=b"\n", qual=b"I"):
    return np.frombuffer(b"".join(b"@r%d x" % i + eol + r + eol + b"+" + eol + qual * len(r) + eol for i, r in enumerate(reads)),
                         dtype=np.uint8)


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("read_len", [150, 31, 40, 1000])
def test_radix_records_of_one_length_take_packed_tiles(kmm, syn, oracle, eol, read_len):
    """Raw FASTQ whose reads all have one length: the chunk is compacted into flat reads on the device
    (k_rec_scatter), found uniform (k_rec_uniform) and mapped through pass 1's packed tiles; same counts as the
    oracle on the reads themselves (command_line_interface.py:102-111 + mapper.pyx:53-69), for both line endings,
    whole and cut at arbitrary bytes."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(6000, seed=371)
    mx = index.max_node_id()
    n_reads = 7001
    bases, offs = syn.make_reads(genome, n_reads, read_len, seed=372)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    raw = _fastq([bases[offs[i]:offs[i + 1]].tobytes() for i in range(n_reads)], eol, qual=b"@")
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("part_shift", 6)
        dev.set_param("host_pack_threads", 0)          # (this test is about the DEVICE-side compaction; test_gpu_hostpack.py has the host's)
        used, n_rec = dev.map_records(raw, fmt=_lib.FORMAT_FASTQ)
        assert (used, n_rec) == (raw.shape[0], n_reads)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        pos, total = 0, 0
        while pos < raw.shape[0]:
            used, n_rec = dev.map_records(np.ascontiguousarray(raw[pos:pos + 333337]), fmt=_lib.FORMAT_FASTQ)
            assert used > 0
            pos += used
            total += n_rec
        assert total == n_reads
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("radix_batches") >= 2 and dev.get_param("direct_batches") == 0


def test_radix_records_from_a_device_buffer_at_any_byte_offset(kmm, syn, oracle):
    """A raw chunk that already lies in HBM and starts at an address that is no multiple of 16 (the second piece of a
    call beyond 2^30 bytes starts where the first one's last record ended): the compaction kernels load 16 bytes per
    lane from unaligned addresses.  Same counts as the oracle for every offset 0..17."""
    import torch
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(5000, seed=391)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 4000, 20, 200, seed=392)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31)
    raw = _fastq([bases[offs[i]:offs[i + 1]].tobytes() for i in range(4000)])
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        for shift in (0, 1, 3, 4, 7, 8, 15, 16, 17):
            buf = torch.zeros(raw.shape[0] + 64, dtype=torch.uint8, device="cuda")
            buf[shift:shift + raw.shape[0]] = torch.from_numpy(raw.copy()).cuda()
            view = buf[shift:shift + raw.shape[0]]
            torch.cuda.synchronize()      # (torch's copy runs on torch's stream, the map call on the handle's)
            dev.reset()
            used, n_rec = dev.map_records(view, fmt=_lib.FORMAT_FASTQ)
            assert (used, n_rec) == (raw.shape[0], 4000), shift
            assert np.array_equal(dev.get_node_counts(), expect), shift


def test_radix_records_edge_cases(kmm, syn, oracle):
    """Compaction corner cases: empty sequence lines, reads shorter than k, a read that ends exactly at a 16-byte lane /
    1024-byte tile boundary, headers and quality lines full of newline-free junk, FASTA2, and the error reports (raw
    byte offsets, as on the direct path)."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(3000, k=5, seed=381, plant=False)
    mx = index.max_node_id()
    g = syn.ACGT[genome]
    rng = np.random.default_rng(382)
    reads, pos = [], 0
    for i in range(4000):
        n = int(rng.choice([0, 1, 4, 5, 6, 11, 12, 16, 27, 1020, 1024, 1019, 300]))
        reads.append(g[pos:pos + n].tobytes())
        pos = (pos + n + 7) % (len(g) - 2000)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 5)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 0)          # (the device-side compaction's corner cases)
        for fmt, raw in ((_lib.FORMAT_FASTQ, _fastq(reads)),
                         (_lib.FORMAT_FASTA2, np.frombuffer(b"".join(b">h%d\n" % i + r + b"\n" for i, r in enumerate(reads)), dtype=np.uint8))):
            dev.reset()
            used, n_rec = dev.map_records(raw, fmt=fmt, k=5)
            assert (used, n_rec) == (raw.shape[0], len(reads))
            assert np.array_equal(dev.get_node_counts(), expect)
            # the last record incomplete: left to the caller, its sequence line is not mapped
            dev.reset()
            used, n_rec = dev.map_records(np.ascontiguousarray(raw[:-3]), fmt=fmt, k=5)
            assert n_rec == len(reads) - 1 and raw[used - 1] == 10
            e2, _ = oracle.map_reads(index, mx, bases[:offs[-2]], offs[:-1], 5)
            assert np.array_equal(dev.get_node_counts(), e2)
        # a non-nucleotide on a sequence line: reported with its RAW byte offset
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1\nACGTXCGTAC\n+\nIIIIIIIIII\n", dtype=np.uint8), k=5)
        with pytest.raises(ValueError, match="offset 8"):
            dev.get_node_counts()
        dev.reset()
        dev.map_records(np.frombuffer(b"@r1\nACGT\nACGT\nIIII\n", dtype=np.uint8), k=3)
        with pytest.raises(ValueError, match="record structure"):
            dev.get_node_counts()
        dev.reset()
        assert dev.map_records(np.frombuffer(b"@r1\nACGT", dtype=np.uint8), k=3) == (0, 0)
        assert dev.get_node_counts().sum() == 0


@pytest.mark.parametrize("modulo", [16411, 40009, 200_000_033, 452_930_477, 2_147_483_629])
def test_division_edge_values(kmm, oracle, modulo):
    """Pass 1 divides by the modulo with a magic multiply and a 32-bit remainder (fastdiv_m31): values around every
    kind of boundary — multiples of the modulo +-1, 2^32 and 2^52 boundaries, the top of the 64-bit range — must land
    on the same index entries as the oracle's hardware divide (mapper.pyx:54), on both paths."""
    rng = np.random.default_rng(351)
    M = np.uint64(modulo)
    n_mult = rng.integers(0, (2 ** 64 - 1) // modulo, size=3000, dtype=np.uint64)
    edge = np.concatenate([n_mult * M, n_mult * M + np.uint64(1), n_mult * M + (M - np.uint64(1)),
                           np.array([0, 1, modulo - 1, modulo, modulo + 1, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 52 - 1,
                                     2 ** 52, 2 ** 53 + 1, 2 ** 62 - 1, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64),
                           rng.integers(0, 2 ** 64 - 1, size=3000, dtype=np.uint64)])
    from kmer_mapper_amd.kmer_index import KmerIndex
    keys = np.unique(edge[rng.integers(0, len(edge), 4000)])
    index = KmerIndex.from_flat_kmers(keys, np.arange(len(keys), dtype=np.int64), modulo)
    mx = len(keys) - 1
    q = np.concatenate([edge, edge[::3] + np.uint64(1), edge[::5] - np.uint64(1)])
    expect = oracle.map_kmers(index, mx, q)
    assert expect.sum() >= len(keys)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for path in ((2, 1) if dev.get_param("radix_available") == 1 else (1,)):
            dev.set_param("path", path)
            dev.reset()
            dev.map_kmers(q)
            assert np.array_equal(dev.get_node_counts(), expect), path


@pytest.mark.parametrize("skewed", [False, True])
def test_flush_orders_agree(kmm, syn, oracle, skewed):
    """Per-entry counts -> node counts (gpu_counter.py:26-37) through the node-ordered entry list (indexes with few
    entries per node) and in bucket order with LDS aggregation (hot nodes): same vector, also when flushes of both
    kinds alternate on one handle and counts wrap around 2^32."""
    index, genome = syn.make_index(60000, seed=361, skewed=skewed)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 30000, 150, seed=362)
    expect, _ = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        assert dev.get_param("radix_sorted_flush") == (0 if skewed else 1)   # 60 entries per node when skewed
        total = np.zeros_like(expect)
        for rep, order in enumerate((1, 0, 1, 1, 0)):
            dev.set_param("radix_sorted_flush", order)
            dev.map_reads_uniform(bases, 30000, 150, 31)
            total = total + expect                                             # uint32: wraps like the device vector
            assert np.array_equal(dev.get_node_counts(), total), (rep, order)


@pytest.mark.parametrize("read_len", [31, 47, 150, 4095, 4096, 5000, 70001])
def test_uniform_reads_of_any_length(kmm, syn, oracle, read_len):
    """kmm_map_reads_uniform works out every lane's offset inside its read from one division per tile plus 32-bit
    arithmetic per lane (reads shorter than a tile: float reciprocal; longer: one conditional subtraction): read
    lengths on both sides of the 4096-position tile, of k itself, and far beyond a tile."""
    index, genome = syn.make_index(30000, seed=371)
    mx = index.max_node_id()
    n_reads = max(3, 3_000_000 // read_len)
    bases, offs = syn.make_reads(genome, n_reads, read_len, seed=372)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    assert n == n_reads * (read_len - 30)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for path in (2, 1):
            dev.set_param("path", path)
            dev.reset()
            dev.map_reads_uniform(bases, n_reads, read_len, 31)
            assert np.array_equal(dev.get_node_counts(), expect), path


def test_conservation_self_check_is_armed_in_production(kmm, syn, oracle):
    """Every synchronising call compares the k-mers emitted by pass 1, gathered by pass 2 and probed by pass 3 (+ those
    pass 2 dropped as absent) and refuses to return counts when they differ (KMM_ERR_INTERNAL, sticky until
    kmm_reset_counts) — the tripwire for a work item processed twice (the round-2 race).  The
    mismatch is forced through the debug parameter; the normal run before and after it passes the check."""
    from kmer_mapper_amd._lib import KmmError
    index, genome = syn.make_index(30000, seed=811)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 20000, 150, seed=812)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)          # check passes on a normal run
        dev.map_reads(bases, offs, 31)
        dev.set_param("debug_skew_p2_counter", 1024)                  # "one item gathered twice"
        with pytest.raises(KmmError, match="self-check failed"):
            dev.get_node_counts()
        with pytest.raises(KmmError, match="self-check failed"):     # sticky: no counts until the reset
            dev.synchronize()
        dev.reset()
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("radix_p2_kmers") == n == dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped")


@pytest.mark.parametrize("read_len", [33, 62, 101, 151, 251, 1000])
def test_packed_tiles_equal_position_tiles_and_the_oracle(kmm, syn, oracle, read_len):
    """Pass 1 on reads of one length takes tiles of whole reads (kmm_tile.hpp, tile_packed_*): lanes-per-read and
    windows-per-lane geometries from one lane per read up to 61 lanes per read, tiles that start at any byte offset
    (odd lengths), reverse complements, the last tile partly empty, and a device buffer that is not 16-byte aligned
    (byte-wise staging) — against the position-based tiles and the oracle (reference: windows never span reads,
    kmer_mapper/util.py:72)."""
    import torch
    index, genome = syn.make_index(40000, seed=391)
    mx = index.max_node_id()
    n_reads = max(700, 1_500_000 // read_len) + 3
    bases, offs = syn.make_reads(genome, n_reads, read_len, seed=392)
    for rc in (False, True):
        expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=rc, n_threads=4)
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            dev.set_param("path", 2)
            assert dev.get_param("radix_packed_tiles") == 1
            dev.map_reads_uniform(bases, n_reads, read_len, 31, also_revcomp=rc)
            assert np.array_equal(dev.get_node_counts(), expect), ("packed", rc)
            assert dev.get_stats(reset=True)[0] == (2 if rc else 1) * n
            dev.reset()
            dev.set_param("radix_packed_tiles", 0)
            dev.map_reads_uniform(bases, n_reads, read_len, 31, also_revcomp=rc)
            assert np.array_equal(dev.get_node_counts(), expect), ("flat", rc)
            dev.reset()
            dev.set_param("radix_packed_tiles", 1)
            t = torch.empty(bases.shape[0] + 7, dtype=torch.uint8, device="cuda")
            t[7:] = torch.from_numpy(bases).cuda()
            dev.map_reads_uniform(t[7:], n_reads, read_len, 31, also_revcomp=rc)      # base pointer at +7 bytes
            assert np.array_equal(dev.get_node_counts(), expect), ("unaligned", rc)


@pytest.mark.parametrize("fine_bits, per_bit", [(7, 1), (8, 2), (9, 4)])
def test_filter_with_several_buckets_per_bit(kmm, syn, oracle, fine_bits, per_bit):
    """Coarse partitions of 2^19, 2^20 and 2^21 buckets: pass 2 folds 1, 2 or 4 buckets of the index's occupancy bitmap
    into one bit of its 64 KB LDS copy (sparse tables such as the customary modulo 452 930 477 keep their filter)."""
    index, genome = syn.make_index(300000, seed=351, modulo=3000017)
    mx = index.max_node_id()
    bases, offs = syn.make_ragged_reads(genome, 40000, 0, 260, seed=352)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=True, n_threads=4)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("part_shift", 12)
        dev.set_param("fine_bits", fine_bits)
        dev.set_param("path", 2)
        assert dev.get_param("radix_filter_buckets_per_bit") == per_bit
        dev.map_reads(bases, offs, 31, also_revcomp=True)
        assert np.array_equal(dev.get_node_counts(), expect)
        dropped = dev.get_param("radix_p2_dropped")
        assert 0 < dropped < 2 * n
        assert dev.get_param("radix_p3_kmers") + dropped == 2 * n


def test_calls_larger_than_a_sub_batch(kmm, syn, oracle):
    """A map call with more k-mer slots than `radix_sub_batch_kmers` (default 2^32 - 2 blocks: configs[4]'s batches beyond
    ~35 M reads; halved by the library when HBM is short) is cut into equal sub-batches, each a full run of the three
    passes.  Forced here with the smallest cap (4 blocks of 8192 slots = some eighty sub-batches per call) on every
    entry point of the radix path: whole-read tiles, position tiles, ragged reads, the k-mer operator, reverse
    complements, raw records — the counts are those of the oracle (mapper.pyx:53-69) and of one uncut call."""
    from kmer_mapper_amd import _lib
    index, genome = syn.make_index(40_000, seed=901)
    mx = index.max_node_id()
    R, L, k = 20_000, 150, 31
    bases, offs = syn.make_reads(genome, R, L, seed=902)
    expect, _ = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
    expect_rc, _ = oracle.map_reads(index, mx, bases, offs, k, also_revcomp=True, n_threads=4)
    rng = np.random.default_rng(903)
    lens = rng.integers(0, 400, size=6000)
    roffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rbases = np.ascontiguousarray(np.tile(syn.ACGT[genome], 8)[: int(roffs[-1])])
    expect_ragged, _ = oracle.map_reads(index, mx, rbases, roffs, k, n_threads=4)
    kmers = oracle.extract(bases, offs, k)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        default_cap = dev.get_param("radix_sub_batch_kmers")
        assert default_cap == 2 ** 32 - 2 * 8192
        for bad in (0, 8192, 2 ** 32):
            with pytest.raises(ValueError):
                dev.set_param("radix_sub_batch_kmers", bad)
        for cap in (4 * 8192, 5 * 8192 + 17, default_cap):
            dev.set_param("radix_sub_batch_kmers", cap)
            assert dev.get_param("radix_sub_batch_kmers") == cap
            for packed in (1, 0):
                dev.set_param("radix_packed_tiles", packed)
                dev.reset()
                dev.map_reads_uniform(bases, R, L, k)
                assert np.array_equal(dev.get_node_counts(), expect), (cap, packed)
            dev.set_param("radix_packed_tiles", 1)
            dev.reset()
            dev.map_reads_uniform(bases, R, L, k, also_revcomp=True)
            assert np.array_equal(dev.get_node_counts(), expect_rc), cap
            dev.reset()
            dev.map_reads(rbases, roffs, k)
            assert np.array_equal(dev.get_node_counts(), expect_ragged), cap
            dev.reset()
            dev.map_kmers(kmers)
            assert np.array_equal(dev.get_node_counts(), expect), cap
            dev.reset()
            dev.get_stats(reset=True)
            raw = _fastq([bases[offs[i]:offs[i + 1]].tobytes() for i in range(R)])
            assert dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=k) == (raw.shape[0], R)
            assert np.array_equal(dev.get_node_counts(), expect), cap
            lookups, hits = dev.get_stats(reset=True)
            assert hits == int(expect.astype(np.uint64).sum())


def test_tuning_knobs_change_no_result(kmm, syn, oracle):
    """Every tuning knob of include/kmm.h that only moves work around — grids, schedules, the pre-filters of both paths,
    the flush order — leaves the counts exactly the oracle's (mapper.pyx:53-69); out-of-range values are refused and
    unknown names too."""
    index, genome = syn.make_index(30_000, seed=951)
    mx = index.max_node_id()
    R, L, k = 30_000, 150, 31
    bases, offs = syn.make_reads(genome, R, L, seed=952)
    expect, _ = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
    knobs = [("path", 1, "grid_per_cu", (1, 3, 64)), ("path", 1, "dynamic_schedule", (0, 1)), ("path", 1, "dyn_chunk", (1, 7, 4096)),
             ("path", 1, "occupancy_filter", (0, 1)),
             ("path", 2, "radix_grid_per_cu", (1, 2)), ("path", 2, "radix_filter", (0, 1)), ("path", 2, "radix_sorted_flush", (0, 1)),
             ("path", 2, "radix_packed_tiles", (0, 1)), ("path", 2, "comm_overlap_slices", (1, 64))]
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        for _, path, name, values in knobs:
            dev.set_param("path", path)
            before = dev.get_param(name) if name not in ("dyn_chunk",) else None
            for v in values:
                dev.set_param(name, v)
                dev.reset()
                dev.map_reads_uniform(bases, R, L, k)
                dev.map_reads(bases, offs, k)
                assert np.array_equal(dev.get_node_counts(), 2 * expect), (name, v)
            if before is not None:
                dev.set_param(name, before)
        for name, bad in (("grid_per_cu", 0), ("grid_per_cu", 5000), ("dyn_chunk", 0), ("radix_grid_per_cu", 3),
                          ("comm_overlap_slices", 0), ("part_shift", 14), ("no_such_knob", 1)):
            with pytest.raises(ValueError):
                dev.set_param(name, bad)
        with pytest.raises(ValueError):
            dev.get_param("no_such_knob")
        for name in ("radix_view_bytes", "direct_view_bytes"):
            assert dev.get_param(name) > 0


def test_reads_packed_on_the_host_before_they_cross_pcie(kmm, syn, oracle):
    """`host_pack_threads` > 0: reads of one length in host memory (default table, a batch that takes the radix path) are
    packed to 2 bits per base by host threads (csrc/kmm_hostpack.hpp) and mapped by pass 1's 2-bit front end — the
    counts are the oracle's (extraction as util.py:71-75 with N -> A, command_line_interface.py:41; lookup
    mapper.pyx:53-69), for read lengths that are no multiple of 4, lower case and N, reverse complements, ragged reads
    with their offsets, and across calls; a byte outside the table sends the call down the ordinary route, which reports it; device buffers, custom
    tables and small batches never take the packed route."""
    index, genome = syn.make_index(50_000, seed=971)
    mx = index.max_node_id()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("host_pack_threads", 3)
        assert dev.get_param("host_pack_threads") == 3
        n_packed = 0
        for R, L, k in ((40_000, 150, 31), (30_001, 151, 31), (70_000, 37, 21), (9_999, 1003, 31)):
            bases, offs = syn.make_reads(genome, R, L, seed=972 + L)          # (lower case, N, substitutions)
            for rc in (False, True):
                expect, _ = oracle.map_reads(index, mx, bases, offs, k, also_revcomp=rc, n_threads=4)
                dev.reset()
                dev.map_reads_uniform(bases, R, L, k, also_revcomp=rc)
                dev.map_reads_uniform(bases, R, L, k, also_revcomp=rc)         # (the page-locked buffer is reused)
                assert np.array_equal(dev.get_node_counts(), 2 * expect), (R, L, k, rc)
                n_packed += 2
                assert dev.get_param("host_packed_calls") == n_packed
        # the page-locked staging ring wrapped many times (8 slots of 4 KiB / 64 KiB instead of 16 MiB): same counts
        R, L, k = 30_001, 151, 31
        bases, offs = syn.make_reads(genome, R, L, seed=972 + L)
        expect, _ = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
        for slot_kb in (4, 64):
            dev.set_param("debug_ring_slot_kb", slot_kb)
            dev.reset()
            dev.map_reads_uniform(bases, R, L, k)
            assert np.array_equal(dev.get_node_counts(), expect), slot_kb
            n_packed += 1
            assert dev.get_param("host_packed_calls") == n_packed
        broken = bases.copy()
        broken[len(broken) // 2 + 3] = ord("!")                               # refused in the middle of a wrapped ring
        dev.reset()
        dev.map_reads_uniform(broken, R, L, k)
        with pytest.raises(ValueError, match="offset %d" % (len(broken) // 2 + 3)):
            dev.get_node_counts()
        dev.set_param("debug_ring_slot_kb", 0)
        dev.reset()
        # ragged reads (kmm_map_reads with host offsets): the same packed bases + the read-start bitset
        rbases, roffs = syn.make_ragged_reads(genome, 60_000, 0, 260, seed=975)
        for k in (31, 9):
            expect, _ = oracle.map_reads(index, mx, rbases, roffs, k, n_threads=4)
            dev.reset()
            dev.map_reads(rbases, roffs, k)
            assert np.array_equal(dev.get_node_counts(), expect), k
            n_packed += 1
            assert dev.get_param("host_packed_calls") == n_packed
        bad_offs = roffs.copy()
        bad_offs[1000] = bad_offs[999] - 1 if bad_offs[999] > 0 else bad_offs[1001] + 5
        dev.reset()
        dev.map_reads(rbases, bad_offs, 31)
        with pytest.raises(ValueError, match="non-decreasing"):
            dev.get_node_counts()
        n_packed += 1
        dev.reset()
        # a byte that is no nucleotide: the ordinary route maps the call and reports the byte's offset
        R, L, k = 40_000, 150, 31
        bases, offs = syn.make_reads(genome, R, L, seed=980)
        broken = bases.copy()
        broken[123_457] = ord("X")
        dev.reset()
        dev.map_reads_uniform(broken, R, L, k)
        with pytest.raises(ValueError, match="offset 123457"):
            dev.get_node_counts()
        assert dev.get_param("host_packed_calls") == n_packed
        dev.reset()
        expect, _ = oracle.map_reads(index, mx, bases, offs, k, n_threads=4)
        # not taken: a caller's table, a batch on the direct path
        lut = np.full(256, 0xFF, dtype=np.uint8)
        for i, c in enumerate(b"ACGT"):
            lut[c] = lut[c + 32] = i
        lut[ord("N")] = lut[ord("n")] = 0
        dev.map_reads_uniform(bases, R, L, k, lut=lut)
        assert np.array_equal(dev.get_node_counts(), expect)
        dev.reset()
        dev.set_param("path", 1)
        dev.map_reads_uniform(bases, R, L, k)
        assert np.array_equal(dev.get_node_counts(), expect)
        assert dev.get_param("host_packed_calls") == n_packed
        with pytest.raises(ValueError):
            dev.set_param("host_pack_threads", -1)
