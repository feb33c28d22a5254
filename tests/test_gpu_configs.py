"""GPU parity at BASELINE.json's real index sizes: configs[1] (10 M-k-mer index) and configs[2] (100 M-k-mer
index), each with a full 10 M-read batch.  At these sizes the oracle cannot check every read in seconds, so the
full batch is checked through size-independent properties (every entry point and both paths agree bit for
bit, linearity over splits, hit-rate window) and a 200 k-read sample is checked against the oracle on the
probe flavour each size really selects (Bloom filter at 10 M, wide buckets at 100 M, radix path for the batch).
Semantics matched: reference kmer_mapper/mapper.pyx:53-69."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kmm():
    from kmer_mapper_amd import _lib
    assert _lib.device_count() >= 1, "GPU tests need a HIP device"
    import kmer_mapper_amd.engine as engine
    return engine


@pytest.mark.parametrize("n_index", [10_000_000, 100_000_000], ids=["configs1_10M_index", "configs2_100M_index"])
def test_full_batch_at_config_size(kmm, oracle, n_index):
    import torch
    from kmer_mapper_amd import synthetic as syn
    R, L, k = 10_000_000, 150, 31
    index, genome = syn.make_index(n_index, k=k, seed=1, gpu_builder=True)      # kmm_build_index
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=1001)
    offs = torch.arange(R + 1, dtype=torch.int64, device="cuda") * L
    del g_ascii
    torch.cuda.synchronize()
    n_kmers = R * (L - k + 1)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        # the layout this size selects by itself
        if n_index == 10_000_000:
            assert dev.get_param("bloom_filter_bytes") > 0 and dev.get_param("wide_buckets") == 0
        else:
            assert dev.get_param("wide_buckets") == 1 and dev.get_param("occupancy_filter") == 0
        assert dev.get_param("radix_available") == 1
        assert dev.get_param("radix_sorted_flush") == 1     # one entry per node: the node-ordered entry list is built
        dev.set_timing(True)
        res = {}
        for name, path, general in (("direct_uniform", 1, False), ("direct_general", 1, True),
                                    ("radix_uniform", 2, False), ("auto_uniform", 0, False)):
            dev.reset()
            dev.set_param("path", path)
            if general:
                dev.map_reads(reads, offs, k)
            else:
                dev.map_reads_uniform(reads, R, L, k)
            res[name] = dev.get_node_counts()
            if path != 1:      # radix path: conservation of k-mers through the passes
                assert dev.get_param("radix_p2_kmers") == n_kmers and dev.get_param("radix_p3_kmers") == n_kmers, name
            assert dev.get_stats(reset=True)[0] == n_kmers, name
        t = dev.get_timing()
        assert t["k_rx_p1"][1] == 2, "a 10 M-read batch takes the radix path by itself (auto)"
        for name in res:
            assert np.array_equal(res[name], res["direct_uniform"]), name
        hits = int(res["direct_uniform"].astype(np.uint64).sum())
        assert 0.15 < hits / n_kmers < 0.25                      # ~0.18-0.20
        # linearity over splits: three unequal pieces, alternating paths, accumulate to the same vector
        dev.reset()
        cuts = [0, 3_333_333, 3_333_334, R]
        for i in range(3):
            dev.set_param("path", 2 if i != 1 else 1)
            a, b = cuts[i], cuts[i + 1]
            dev.map_reads_uniform(reads[a * L:b * L], b - a, L, k)
        assert np.array_equal(dev.get_node_counts(), res["direct_uniform"])
        # 200 k-read sample against the oracle, on every path
        n_s = 200_000
        sample = reads[:n_s * L].cpu().numpy()
        s_offs = np.arange(n_s + 1, dtype=np.int64) * L
        expect, n = oracle.map_reads(index, mx, sample, s_offs, k, n_threads=16)
        assert n == n_s * (L - k + 1)
        for path in (0, 1, 2):
            dev.reset()
            dev.set_param("path", path)
            dev.map_reads_uniform(reads[:n_s * L], n_s, L, k)
            assert np.array_equal(dev.get_node_counts(), expect), path
        # operator path (map_kmers_to_graph_index drop-in) on the sample's k-mers, membership as well
        km = kmm.extract_kmers(sample, s_offs, k)
        assert np.array_equal(km, oracle.extract(sample, s_offs, k))
        for path in (1, 2):
            dev.reset()
            dev.set_param("path", path)
            dev.map_kmers(km)
            assert np.array_equal(dev.get_node_counts(), expect), path
        assert np.array_equal(dev.in_index(km[:2_000_000]), oracle.in_index(index, km[:2_000_000]))


def test_customary_large_modulo_uses_8192_bucket_slices(kmm, oracle):
    """graph_kmer_index's customary modulo 452 930 477 ([UPSTREAM-UNVERIFIED]) needs more than
    256 x 256 slices of 4096 buckets: the radix path then runs with 8192-bucket slices."""
    from kmer_mapper_amd import synthetic as syn
    index, genome = syn.make_index(2_000_000, k=31, seed=11, modulo=452_930_477, gpu_builder=True)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 300_000, 150, seed=12)
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, n_threads=16)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        assert dev.get_param("radix_available") == 1 and dev.get_param("part_shift") == 13
        for path in (1, 2):
            dev.reset()
            dev.set_param("path", path)
            dev.map_reads_uniform(bases, 300_000, 150, 31)
            assert np.array_equal(dev.get_node_counts(), expect), path
        assert dev.get_param("radix_p3_kmers") == n
