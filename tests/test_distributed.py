"""world_size-2 gloo tests (CPU) of the N>1 path: read sharding + the single sum-reduce of the
uint32 count vectors.  The per-rank map step is played by the oracle here (tests may use it); on
GPUs the same code runs with the HIP engine and backend nccl (= RCCL)."""
import os
import socket

import numpy as np
import pytest

from kmer_mapper_amd.distributed import as_int32_bits, chunk_owner, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 10, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    assert [chunk_owner(i, 4) for i in range(6)] == [0, 1, 2, 3, 0, 1]


def test_int32_view_wraps_like_uint32():
    a = np.array([0xFFFFFFFF, 5], dtype=np.uint32)
    b = np.array([2, 0xFFFFFFFE], dtype=np.uint32)
    s = (as_int32_bits(a) + as_int32_bits(b)).view(np.uint32)
    assert s.tolist() == [1, 3]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmer_mapper_amd import synthetic as syn
        from kmer_mapper_amd.distributed import reduce_node_counts, shard_range
        from oracle import oracle
        index, genome = syn.make_index(500, seed=7)
        mx = index.max_node_id()
        bases, offs = syn.make_reads(genome, 1001, 150, seed=8)
        lo, hi = shard_range(1001, rank, world)
        mine, _ = oracle.map_reads(index, mx, bases[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo], 31)
        mine[3] += np.uint32(0xFFFFFFF0)          # force a wrap-around in the sum
        t = torch.from_numpy(mine.view(np.int32).copy())
        reduce_node_counts(t, dst=0)
        t2 = torch.from_numpy(mine.view(np.int32).copy())
        reduce_node_counts(t2, all_ranks=True)
        whole, _ = oracle.map_reads(index, mx, bases, offs, 31)
        whole[3] += np.uint32((0xFFFFFFF0 * world) & 0xFFFFFFFF)
        ok_all = np.array_equal(t2.numpy().view(np.uint32), whole)
        ok_root = rank != 0 or np.array_equal(t.numpy().view(np.uint32), whole)
        q.put((rank, bool(ok_all and ok_root)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_reduce_is_bit_exact():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


class _NoRccl:
    """Stands for a DeviceIndex on a host where the RCCL library cannot be loaded: no id on rank 0, and nobody may reach
    comm_init (a collective the other ranks would then wait in for ever)."""
    device = 0

    @staticmethod
    def comm_unique_id():
        raise RuntimeError("librccl.so could not be loaded")

    def comm_init(self, unique_id, n_ranks, rank):  # pragma: no cover - reaching it is the failure
        raise AssertionError("comm_init was reached without an id")


def _worker_no_id(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmer_mapper_amd.distributed import init_rccl_comm
        try:
            init_rccl_comm(_NoRccl())
            q.put((rank, "joined"))
        except RuntimeError as exc:
            q.put((rank, "refused: %s" % exc))
        dist.barrier()                            # every rank is still in step with the others afterwards
    finally:
        dist.destroy_process_group()


def test_every_rank_refuses_together_when_rank_0_has_no_unique_id():
    """bench.py / map_bnp fall back to torch.distributed when the library's communicator does not come up; that decision
    must be the same on every rank, or one rank sits in a broadcast the others never enter."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_no_id, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0].startswith("refused") and "librccl" in res[0]
    assert res[1].startswith("refused")
