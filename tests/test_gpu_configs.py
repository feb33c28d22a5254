"""GPU parity at BASELINE.json's real index sizes: configs[1] (10 M-k-mer index) and configs[2] (100 M-k-mer
index), each with a full 10 M-read batch.  At these sizes the oracle cannot check every read in seconds, so the
full batch is checked through size-independent properties (every entry point and both paths agree bit for
bit, linearity over splits, hit-rate window) and a 200 k-read sample is checked against the oracle on the
probe flavour each size really selects (Bloom filter at 10 M, wide buckets at 100 M, radix path for the batch).
Semantics matched: reference kmer_mapper/mapper.pyx:53-69."""
import os

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
                assert dev.get_param("radix_p2_kmers") == n_kmers, name
                assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == n_kmers, name
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
        assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == n


def test_modulo_above_2_pow_29_takes_the_radix_path(kmm, oracle):
    """a table with more than 256 x 256 slices of 8192 buckets (modulo > 2^29; the reference's int32
    tables allow every modulo below 2^31, mapper.pyx:22-23,31-32,53-56) used to fall off the radix path.  3e8 entries,
    modulo ~6e8: the fan-out goes beyond 256 per pass; a full 10 M-read batch through direct == radix == auto,
    conservation counters, and a 200 k-read sample against the oracle.  The direct view of an index this size
    (24 GB) is only packed when the first direct-path batch asks for it."""
    import torch
    from kmer_mapper_amd import synthetic as syn
    R, L, k = 10_000_000, 150, 31
    index, genome = syn.make_index(300_000_000, k=k, seed=1, gpu_builder=True)
    assert index._modulo > 2 ** 29
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=1001)
    del g_ascii, genome
    torch.cuda.synchronize()
    n_kmers = R * (L - k + 1)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        assert dev.get_param("radix_available") == 1 and dev.get_param("radix_unavailable_reason") == 0
        assert max(dev.get_param("n_coarse_partitions"), dev.get_param("n_fine_per_coarse")) > 256
        assert dev.get_param("direct_view_resident") == 0          # HBM budget: one view until the other is needed
        res = {}
        for name, path in (("radix", 2), ("auto", 0), ("direct", 1)):
            dev.reset()
            dev.set_param("path", path)
            dev.map_reads_uniform(reads, R, L, k)
            res[name] = dev.get_node_counts()
            if path != 1:
                assert dev.get_param("radix_p2_kmers") == n_kmers, name
                assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == n_kmers, name
            else:
                assert dev.get_param("direct_view_resident") == 1
            assert dev.get_stats(reset=True)[0] == n_kmers, name
        assert np.array_equal(res["radix"], res["direct"]) and np.array_equal(res["auto"], res["direct"])
        hits = int(res["direct"].astype(np.uint64).sum())
        assert 0.15 < hits / n_kmers < 0.25
        n_s = 200_000
        sample = reads[:n_s * L].cpu().numpy()
        s_offs = np.arange(n_s + 1, dtype=np.int64) * L
        expect, n = oracle.map_reads(index, mx, sample, s_offs, k, n_threads=8)
        assert n == n_s * (L - k + 1)
        for path in (1, 2):
            dev.reset()
            dev.set_param("path", path)
            dev.map_reads_uniform(reads[:n_s * L], n_s, L, k)
            assert np.array_equal(dev.get_node_counts(), expect), path
        km = kmm.extract_kmers(sample[:20_000 * L], s_offs[:20_001], k)
        assert np.array_equal(dev.in_index(km), oracle.in_index(index, km))


def test_configs4_index_on_one_gpu(kmm, oracle):
    """BASELINE configs[4]'s index on one GPU: 10^9 k-mers, modulo 2 000 000 011 — the largest the reference's int32
    tables allow (mapper.pyx:22-23,53-56: `_hashes_to_index`, `_n_kmers` int32, `kmers[i] % modulo`).  Index generated
    and built on the GPU (synthetic.make_index_torch / kmm_build_index); a 12 M-read batch through the radix path
    (477 x 512 slices of 8192 buckets; ONE sub-batch of 1.44e9 k-mers) with the conservation counters; a 200 k-read
    sample on both paths against the oracle run on the host copy of the index; linearity over a split of the batch;
    membership of 20 k k-mers.  Semantics: mapper.pyx:53-69."""
    import torch
    from kmer_mapper_amd import synthetic as syn
    R, L, k = 12_000_000, 150, 31
    index, g_ascii = syn.make_index_torch(1_000_000_000, k=k, seed=1)
    assert index._modulo == 2_000_000_011
    mx = index.max_node_id()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=1001)
    del g_ascii
    torch.cuda.synchronize()
    n_kmers = R * (L - k + 1)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        host = index.to_host()            # the oracle's copy (host numpy arrays); the handle owns its own
        del index
        torch.cuda.empty_cache()
        assert dev.get_param("radix_available") == 1 and dev.get_param("part_shift") == 13
        assert dev.get_param("n_coarse_partitions") * dev.get_param("n_fine_per_coarse") >= dev.get_param("n_partitions") > 200_000
        assert dev.get_param("direct_view_resident") == 0          # 80 GB: packed only when a direct batch asks for it
        dev.set_timing(True)
        dev.set_param("path", 2)
        dev.map_reads_uniform(reads, R, L, k)
        full = dev.get_node_counts()
        assert dev.get_param("radix_p2_kmers") == n_kmers
        assert dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == n_kmers
        assert dev.get_stats(reset=True)[0] == n_kmers
        assert dev.get_timing()["k_rx_p1"][1] == 1, "1.44e9 k-mers are one sub-batch (the index slices are streamed once)"
        hits = int(full.astype(np.uint64).sum())
        assert 0.15 < hits / n_kmers < 0.25
        # linearity: two unequal pieces accumulate to the same vector
        dev.reset()
        cut = 4_999_999
        dev.map_reads_uniform(reads[:cut * L], cut, L, k)
        dev.map_reads_uniform(reads[cut * L:], R - cut, L, k)
        assert np.array_equal(dev.get_node_counts(), full)
        del full
        # 200 k-read sample against the oracle: radix path, then the direct path (packs the direct view on first use)
        n_s = 200_000
        sample = reads[:n_s * L].cpu().numpy()
        s_offs = np.arange(n_s + 1, dtype=np.int64) * L
        expect, n = oracle.map_reads(host, mx, sample, s_offs, k, n_threads=6)
        assert n == n_s * (L - k + 1)
        dev.reset()
        dev.map_reads_uniform(reads[:n_s * L], n_s, L, k)
        assert np.array_equal(dev.get_node_counts(), expect), "radix"
        del reads
        torch.cuda.empty_cache()
        dev.reset()
        dev.set_param("path", 1)
        dev.map_reads_uniform(sample, n_s, L, k)
        assert np.array_equal(dev.get_node_counts(), expect), "direct"
        assert dev.get_param("direct_view_resident") == 1
        km = kmm.extract_kmers(sample[:20_000 * L], s_offs[:20_001], k)
        assert np.array_equal(dev.in_index(km), oracle.in_index(host, km))


def test_sub_batch_cap_is_halved_when_the_buffers_do_not_fit(kmm, oracle):
    """The out-of-memory route of the radix path's buffer sizing: when the buffers of a sub-batch at the current cap
    cannot be allocated the call takes one sub-batch more until it is cut finely enough (the caller's parameter stays) — here
    forced with a test hook that treats a pass-1 buffer beyond 1.6 GB as out of memory: a 2.5 M-read call (3.0e8 k-mer
    slots, 2.4 GB at one sub-batch) ends up as two sub-batches at a cap of 2^28.  Same counts as the direct path and,
    on a sample, as the oracle (mapper.pyx:53-69); below the floor of 2^28 slots the call fails with MemoryError."""
    import torch
    from kmer_mapper_amd import synthetic as syn
    R, L, k = 2_500_000, 150, 31
    index, genome = syn.make_index(1_000_000, k=k, seed=5, gpu_builder=True)
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=77)
    del g_ascii
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 1)
        dev.map_reads_uniform(reads, R, L, k)
        direct = dev.get_node_counts()
        dev.reset()
        dev.set_param("path", 2)
        dev.set_param("debug_rx_buffer_limit", 1_600_000_000)
        for packed in (1, 0):
            dev.set_param("radix_packed_tiles", packed)
            dev.set_param("radix_sub_batch_kmers", 2 ** 32 - 2 * 8192)
            dev.reset()
            dev.set_timing(True)
            dev.map_reads_uniform(reads, R, L, k)
            assert np.array_equal(dev.get_node_counts(), direct), packed
            assert dev.get_param("radix_sub_batch_kmers") == 2 ** 32 - 2 * 8192                # the caller's value stays
            assert dev.get_param("radix_sub_batch_kmers_effective") == 2 ** 28                 # one sub-batch more (the floor)
            assert dev.get_timing()["k_rx_p1"][1] == 2, "two sub-batches"
            dev.set_timing(False)
            assert dev.get_param("radix_p2_kmers") == R * (L - k + 1)
            dev.get_stats(reset=True)
        dev.set_param("debug_rx_buffer_limit", 100_000_000)         # nothing at or above the floor fits
        dev.reset()
        with pytest.raises(MemoryError):
            dev.map_reads_uniform(reads, R, L, k)
        dev.set_param("debug_rx_buffer_limit", 0)
        dev.reset()
        dev.map_reads_uniform(reads, R, L, k)
        assert np.array_equal(dev.get_node_counts(), direct)
    n_s = 50_000
    sample = reads[: n_s * L].cpu().numpy()
    expect, _ = oracle.map_reads(index, mx, sample, np.arange(n_s + 1, dtype=np.int64) * L, k, n_threads=8)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("path", 2)
        dev.set_param("radix_sub_batch_kmers", 4 * 8192)
        dev.map_reads_uniform(sample, n_s, L, k)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_back_to_back_record_calls_at_configs2_size_keep_every_kmer(kmm, oracle):
    """Round-4 fault, kept as a test: with the compaction kernels of call i + 1 running BESIDE the radix passes of call i,
    pass 1's counting sort lost 100-600 of 1.2e9 k-mers per call in most rounds (a counter cleared by a slow wavefront
    after a faster one had already ranked into it for the next block: rx_sort_emit, fixed; the conservation self-check
    caught every occurrence; profiles/r04/records_overlap_fault.txt).  Six rounds of two 3 GB raw FASTQ calls against the
    100 M index, then the same with the second call's compaction forced onto the copy stream (the constellation that
    exposed the race): every synchronising call passes the self-check, all rounds give one count vector, the flat
    reads give the same vector, and a 100 k-read sample equals the oracle's."""
    import torch
    from kmer_mapper_amd import _lib, synthetic as syn
    R, L, k = 10_000_000, 150, 31
    index, genome = syn.make_index(100_000_000, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    g = torch.from_numpy(syn.ACGT[genome]).cuda()
    fq, reads0 = [], None
    for b in range(2):
        reads = syn.make_reads_torch(g, R, L, seed=1000 + b)
        rec = torch.empty((R, 4 + L + 3 + L + 1), dtype=torch.uint8, device="cuda")
        rec[:, 0:4] = torch.tensor(list(b"@rd\n"), dtype=torch.uint8, device="cuda")
        rec[:, 4:4 + L] = reads.view(R, L)
        rec[:, 4 + L:7 + L] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device="cuda")
        rec[:, 7 + L:7 + 2 * L] = ord("F")
        rec[:, -1] = 10
        fq.append(rec.reshape(-1))
        if b == 0:
            reads0 = reads
        del rec
    del g
    torch.cuda.synchronize()
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        first = None
        for r in range(6):
            dev.reset()
            for b in range(2):
                used, n_rec = dev.map_records(fq[b], fmt=_lib.FORMAT_FASTQ, k=k)
                assert (used, n_rec) == (fq[b].numel(), R)
            got = dev.get_node_counts()                 # (runs the conservation self-check: KMM_ERR_INTERNAL if it fails)
            assert dev.get_param("radix_p2_kmers") == 2 * R * (L - k + 1)
            assert dev.get_stats(reset=True)[0] == 2 * R * (L - k + 1)     # (also clears the passes' counters for the next round)
            if first is None:
                first = got
            assert np.array_equal(got, first), r
        assert dev.get_param("direct_batches") == 0
        # foreign wavefronts beside pass 1: the second call only runs its compaction, on the copy stream, while the first
        # call's passes are under way (tools/records_overlap_bisect.py); the first call's counts must be those of one call
        dev.reset()
        dev.map_records(fq[0], fmt=_lib.FORMAT_FASTQ, k=k)
        single = dev.get_node_counts()
        dev.get_stats(reset=True)
        for r in range(6):
            dev.reset()
            dev.set_param("debug_records_copy_stream", 0)
            dev.map_records(fq[0], fmt=_lib.FORMAT_FASTQ, k=k)
            dev.set_param("debug_records_copy_stream", 1)          # (maps nothing: its kernels only keep the CUs company)
            dev.map_records(fq[1], fmt=_lib.FORMAT_FASTQ, k=k)
            dev.set_param("debug_records_copy_stream", 0)
            assert np.array_equal(dev.get_node_counts(), single), r
            assert dev.get_stats(reset=True)[0] == R * (L - k + 1)
        # the same reads as flat uniform batches give the same vector; a sample of batch 0 equals the oracle
        dev.reset()
        views = [fq[b].view(R, 4 + L + 3 + L + 1)[:, 4:4 + L].contiguous().view(-1) for b in range(2)]
        torch.cuda.synchronize()          # (torch's copy kernels run on torch's stream, the map calls on the handle's)
        for b in range(2):
            dev.map_reads_uniform(views[b], R, L, k)          # (asynchronous: the buffers stay alive until the sync below)
        assert np.array_equal(dev.get_node_counts(), first)
        del views
        n_s = 100_000
        sample = reads0[:n_s * L].cpu().numpy()
        expect, _ = oracle.map_reads(index, mx, sample, np.arange(n_s + 1, dtype=np.int64) * L, k, n_threads=8)
        dev.reset()
        raw = fq[0][: n_s * (4 + L + 3 + L + 1)].contiguous()
        torch.cuda.synchronize()
        assert dev.map_records(raw, fmt=_lib.FORMAT_FASTQ, k=k) == (raw.numel(), n_s)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_direct_view_is_packed_on_first_use_when_deferred(kmm, oracle, monkeypatch):
    """HBM budget of large indexes: with the direct view deferred (forced here on a small index) the radix path
    works without it, the first direct-path batch / kmm_in_index packs it from the radix view's bucket-ordered
    copy, and every result equals the oracle's (also with the L2 pre-filter rebuilt from that copy)."""
    from kmer_mapper_amd import synthetic as syn
    monkeypatch.setenv("KMM_DIRECT_EAGER_BYTES", "0")
    for n_index, modulo in ((30_000, None), (200_000, 452_930_477)):     # Bloom-filter layout; wide layout
        index, genome = syn.make_index(n_index, k=31, seed=71, modulo=modulo, gpu_builder=True)
        mx = index.max_node_id()
        bases, offs = syn.make_ragged_reads(genome, 20_000, 0, 250, seed=72)
        expect, n = oracle.map_reads(index, mx, bases, offs, 31, n_threads=4)
        km = oracle.extract(bases, offs, 31)
        with kmm.DeviceIndex.from_index(index, mx) as dev:
            assert dev.get_param("direct_view_resident") == 0
            dev.set_param("path", 2)
            dev.map_reads(bases, offs, 31)
            assert np.array_equal(dev.get_node_counts(), expect)
            assert dev.get_param("direct_view_resident") == 0
            assert np.array_equal(dev.in_index(km[:50_000]), oracle.in_index(index, km[:50_000]))
            assert dev.get_param("direct_view_resident") == 1
            dev.reset()
            dev.set_param("path", 1)
            dev.map_reads(bases, offs, 31)
            assert np.array_equal(dev.get_node_counts(), expect)
            dev.reset()
            dev.map_kmers(km)
            assert np.array_equal(dev.get_node_counts(), expect)


def test_overlapping_buckets_are_served_by_the_direct_path(kmm, oracle):
    """The reference's loop only follows (hashes_to_index[h], n_kmers[h]) (mapper.pyx:55-58): two buckets may share
    entries.  Such an index has no bucket-ordered copy of bounded size, so the radix view is not built (reason 4) and
    the direct path answers — same counts as the oracle."""
    import types
    from kmer_mapper_amd import synthetic as syn
    index, genome = syn.make_index(5_000, k=31, seed=91)
    h2i, nk = index._hashes_to_index.copy(), index._n_kmers.copy()
    empty = np.flatnonzero(nk == 0)[:200]
    full = np.flatnonzero(nk > 0)[:200]
    h2i[empty], nk[empty] = h2i[full], nk[full]        # 200 empty buckets alias 200 occupied ones (never matched: wrong hash)
    dup = types.SimpleNamespace(_hashes_to_index=h2i, _n_kmers=nk, _nodes=index._nodes, _kmers=index._kmers,
                                _frequencies=index._frequencies, _modulo=index._modulo)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, 5_000, 150, seed=92)
    expect, _ = oracle.map_reads(dup, mx, bases, offs, 31, n_threads=2)
    with kmm.DeviceIndex.from_index(dup, mx) as dev:
        assert dev.get_param("radix_available") == 0 and dev.get_param("radix_unavailable_reason") == 4
        dev.map_reads(bases, offs, 31)
        assert np.array_equal(dev.get_node_counts(), expect)


def test_cli_accumulates_chunks_into_radix_batches_and_records_beyond_2_pow_30(kmm, tmp_path, caplog):
    """`kmer_mapper map` with the reference's default -c (2.5 MB chunks,
    command_line_interface.py:169) accumulates its chunks into GPU batches large enough for the radix path
    (`path_taken: radix` in the log) and gives the counts of one big uniform map call; kmm_map_records takes a raw
    chunk beyond 2^30 bytes (mapped piece by piece inside the library)."""
    import argparse
    import logging
    import torch
    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.command_line_interface import map_bnp
    from kmer_mapper_amd import _lib
    R, L, k = 4_000_000, 150, 31
    index, genome = syn.make_index(100_000_000, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=77)
    del g_ascii, genome
    rec_len = 4 + L + 3 + L + 1                                     # "@rd\n" seq "\n+\n" qual "\n" = 308 bytes
    rec = torch.empty((R, rec_len), dtype=torch.uint8, device="cuda")
    rec[:, 0:4] = torch.tensor(list(b"@rd\n"), dtype=torch.uint8, device="cuda")
    rec[:, 4:4 + L] = reads.view(R, L)
    rec[:, 4 + L:4 + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device="cuda")
    rec[:, 4 + L + 3:4 + L + 3 + L] = ord("F")
    rec[:, -1] = 10
    raw = rec.reshape(-1)                                           # 1.23 GB of FASTQ > 2^30
    assert raw.numel() > 2 ** 30
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        dev.map_reads_uniform(reads, R, L, k)
        expect = dev.get_node_counts()
        dev.reset()
        used, n_rec = dev.map_records(raw, raw.numel(), _lib.FORMAT_FASTQ, k)
        assert (used, n_rec) == (raw.numel(), R)
        assert np.array_equal(dev.get_node_counts(), expect)
    fq = str(tmp_path / "reads.fq")
    raw.cpu().numpy().tofile(fq)
    del raw, rec, reads
    args = argparse.Namespace(kmer_index=index, index_bundle=None, reads=fq, kmer_size=k, n_threads=16, chunk_size=2_500_000,
                              output_file=None, debug=None, max_hits_per_kmer=1000, gpu=True, gpu_hash_map_size=0,
                              map_reverse_complements=False, apply_max_hits_per_kmer=False, host_parser=False, device=0)
    with caplog.at_level(logging.INFO):
        got = map_bnp(args)
    assert np.array_equal(got, expect)
    log = caplog.text
    assert "are accumulated into GPU batches" in log and "path_taken: radix" in log, log[-600:]


def test_a_gigabyte_of_bgzf_maps_like_the_raw_bytes_whatever_the_windows(kmm, oracle, tmp_path):
    """Size-independent property at a configs-sized input (3 M reads = 1 GB of FASTQ, 10 M-k-mer index): the node counts do
    not depend on the route the bytes take — BGZF members inflated on the GPU in ONE call, in windows announced ahead
    (staged under the kernel of the window before), as three ranks' member ranges — or the raw FASTQ packed by the host
    threads (kmm_map_records); every read is counted once; a 40 000-read prefix equals the oracle (mapper.pyx:53-69 on
    util.py:71-75's k-mers)."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    from kmer_mapper_amd import _lib, bgzf_ranges, synthetic as syn
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bgzf_e2e
    n_reads, L, k = 3_000_000, 150, 31
    index, genome = syn.make_index(10_000_000, seed=61, gpu_builder=True)
    mx = index.max_node_id()
    bases, offs = syn.make_reads(genome, n_reads, L, seed=62)
    fq = str(tmp_path / "reads.fq")
    bgzf_e2e.make_fastq(fq, bases, n_reads, L)
    raw = np.fromfile(fq, dtype=np.uint8)
    os.remove(fq)
    view = memoryview(raw)
    with ThreadPoolExecutor(16) as pool:                                   # (zlib releases the GIL; no fork next to a live GPU)
        comp = b"".join(pool.map(lambda p: bgzf_e2e._member(bytes(view[p:p + 0xFF00])), range(0, raw.shape[0], 0xFF00))) + bgzf_e2e._EOF
    buf = np.frombuffer(comp, dtype=np.uint8)
    with kmm.DeviceIndex.from_index(index, mx) as dev:
        used, n_rec = dev.map_records(raw, raw.shape[0], _lib.FORMAT_FASTQ, k)
        assert used == raw.shape[0] and n_rec == n_reads
        want = dev.get_node_counts()
        assert dev.get_param("host_packed_record_calls") >= 1
        # one call
        dev.reset()
        used, n_rec = dev.map_bgzf(buf, fmt=_lib.FORMAT_FASTQ, k=k, first=True, last=True)
        assert used == len(comp) and n_rec == n_reads
        assert np.array_equal(dev.get_node_counts(), want)
        # windows of 100 MB, each announced while the one before is inflated
        dev.reset()
        step, pos, total = 100 << 20, 0, 0
        end = min(step, len(comp))
        before = dev.get_param("bgzf_prestaged_calls")
        while pos < len(comp):
            nxt = min(end + step, len(comp))
            used, n_rec = dev.map_bgzf(buf[pos:end], fmt=_lib.FORMAT_FASTQ, k=k, first=pos == 0, last=end == len(comp),
                                       next_chunk=buf[end:nxt] if nxt > end else None)
            pos += used
            total += n_rec
            if pos < end and end == len(comp):
                continue
            end = nxt
        assert total == n_reads and np.array_equal(dev.get_node_counts(), want)
        assert dev.get_param("bgzf_prestaged_calls") - before == (len(comp) - 1) // step
        # three ranks' member ranges on one handle
        dev.reset()
        total = 0
        for r in range(3):
            lo, s0, hi, s1 = bgzf_ranges.rank_member_range(comp, "fastq", r, 3)
            hi = bgzf_ranges.member_end(comp, hi) if s1 > 0 else hi
            used, n_rec = dev.map_bgzf(buf[lo:hi], fmt=_lib.FORMAT_FASTQ, k=k, first=True, last=True, head_skip=s0,
                                       tail_stop=s1 if s1 > 0 else None)
            assert used == hi - lo
            total += n_rec
        assert total == n_reads and np.array_equal(dev.get_node_counts(), want)
        # and the oracle on a prefix
        n_s = 40_000
        dev.reset()
        dev.map_reads_uniform(bases[:n_s * L], n_s, L, k)
        expect, _ = oracle.map_reads(index, mx, bases[:n_s * L], offs[:n_s + 1], k, n_threads=8)
        assert np.array_equal(dev.get_node_counts(), expect)
