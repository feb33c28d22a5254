"""Multi-GPU data parallelism over read chunks: one process per GPU, index replicated, reads
range-sharded, ONE sum-reduce of the uint32 node-count vectors at the end (RCCL over xGMI on GPUs;
gloo in the CPU tests).  This is the MI355X counterpart of the reference's process pool +
additative_shared_array_map_reduce (kmer_mapper/command_line_interface.py:110-130); the path has no
other exchange step, so no other collective exists.

uint32 counts wrap modulo 2^32 (mapper.pyx:37,68); two's-complement int32 addition produces the
same bits, so the vectors travel as int32 (a dtype every backend reduces) and the result is
bit-exact regardless of reduction order.
"""
import numpy as np


def shard_range(n_items, rank, world_size):
    """Contiguous range [lo, hi) of items (reads or chunks) owned by `rank`; sizes differ by <= 1."""
    n_items, rank, world_size = int(n_items), int(rank), int(world_size)
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_owner(chunk_index, world_size):
    """File-input mode: chunk i goes to rank i mod world_size."""
    return int(chunk_index) % int(world_size)


def as_int32_bits(counts):
    """View a uint32 count vector (numpy array or torch tensor) as int32 without copying."""
    if isinstance(counts, np.ndarray):
        if counts.dtype != np.uint32 and counts.dtype != np.int32:
            raise ValueError("counts must be uint32/int32")
        return counts.view(np.int32)
    import torch
    if counts.dtype == torch.int32:
        return counts
    if counts.element_size() != 4:
        raise ValueError("counts must be a 32-bit integer tensor")
    return counts.view(torch.int32)


def reduce_node_counts(counts, dst=0, group=None, all_ranks=False):
    """Sum the per-rank count vectors.  `counts` is a torch tensor (int32 bits of the uint32
    counts) on the backend's device; after the call rank `dst` (or every rank if all_ranks) holds
    the total.  No-op when torch.distributed is not initialised (single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counts
    t = as_int32_bits(counts)
    if all_ranks:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return counts


def init_rccl_comm(dev, group=None):
    """Join the library's own RCCL communicator (include/kmm.h: kmm_comm_*) with the DeviceIndex `dev`, using an
    initialised torch.distributed group only to hand rank 0's 128-byte unique id to the other ranks.  Afterwards
    dev.comm_reduce_counts() needs no torch."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(dev.comm_unique_id()), dtype=torch.uint8).clone()
    if on_gpu:
        t = t.cuda(dev.device)
    dist.broadcast(t, src=0, group=group)
    dev.comm_init(bytes(t.cpu().numpy().tobytes()), world, rank)
