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
    # 128 bytes of id + one status byte: if rank 0 cannot make an id (RCCL not loadable, ...) it still takes part in the
    # broadcast and EVERY rank raises — a rank 0 that left early would leave the others waiting in a collective nobody joins
    t = torch.zeros(129, dtype=torch.uint8)
    why = ""
    if rank == 0:
        try:
            t[:128] = torch.frombuffer(bytearray(dev.comm_unique_id()), dtype=torch.uint8)
            t[128] = 1
        except Exception as exc:                                        # noqa: BLE001 - reported on every rank below
            why = str(exc)
    if on_gpu:
        t = t.cuda(dev.device)
    dist.broadcast(t, src=0, group=group)
    t = t.cpu()
    if int(t[128]) != 1:
        raise RuntimeError("rank 0 could not create an RCCL unique id" + (": " + why if why else ""))
    dev.comm_init(bytes(t[:128].numpy().tobytes()), world, rank)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_to_gpu_numa_node(device, sysfs="/sys/bus/pci/devices"):
    """One process per GPU: keep this rank's host side — the threads that read, inflate and pack its reads and the page-locked
    buffers they fill — on the NUMA node its GPU hangs off.  The GPU's PCI bus id (kmm_device_pci_bus_id) names
    <sysfs>/<id>/numa_node and local_cpulist; the process's CPU affinity is cut to those CPUs (memory then comes from the
    same node by first touch).  The reference leaves its worker processes unbound (command_line_interface.py:124-130); with
    eight ranks pulling ~55-200 GB/s each from host DRAM the placement decides whether a socket's memory channels or the
    inter-socket links carry it.  Returns {"pci": ..., "numa_node": ..., "cpus": n, "bound": bool, "why": ...};
    KMM_NO_NUMA_BIND=1 leaves the affinity alone."""
    import ctypes
    import os
    from . import _lib
    info = {"pci": None, "numa_node": None, "cpus": None, "bound": False, "why": ""}
    buf = ctypes.create_string_buffer(64)
    try:
        _lib.check(_lib.lib().kmm_device_pci_bus_id(int(device), buf, 64))
    except Exception as exc:                                            # noqa: BLE001 - binding is an optimisation
        info["why"] = "no PCI bus id: %s" % exc
        return info
    info["pci"] = buf.value.decode().lower()
    base = os.path.join(sysfs, info["pci"])
    try:
        node = int(open(os.path.join(base, "numa_node")).read())
        cpus = _parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except (OSError, ValueError) as exc:
        info["why"] = "sysfs: %s" % exc
        return info
    info["numa_node"] = node
    if os.environ.get("KMM_NO_NUMA_BIND") == "1":
        info["why"] = "KMM_NO_NUMA_BIND=1"
        return info
    if node < 0 or not cpus:
        info["why"] = "the platform reports no NUMA node for the device"
        return info
    global _BOUND
    try:
        before = os.sched_getaffinity(0)
        allowed = before & cpus
        if not allowed:
            info["why"] = "none of the node's CPUs is in this process's affinity mask"
            return info
        os.sched_setaffinity(0, allowed)
        info["cpus"] = len(allowed)
        info["bound"] = True
        if _BOUND is None or not (before <= _BOUND["before"]):     # (a second call sees the cut mask: keep the first one's)
            _BOUND = {"before": set(before), "numa_node": node}
        else:
            _BOUND["numa_node"] = node
    except (AttributeError, OSError) as exc:
        info["why"] = "sched_setaffinity: %s" % exc
    return info


_BOUND = None        # {"before": the affinity mask this process had before bind_to_gpu_numa_node cut it, "numa_node": the GPU's}


def packer_cpus_near(page_nodes, sysfs="/sys/devices/system/node", min_share=0.75):
    """Where should the threads that pack a memory-mapped read file run?  bind_to_gpu_numa_node keeps a rank on its GPU's
    node; a file whose page-cache pages lie on the OTHER socket is then read across the socket link — 98 GB/s of FASTQ on 16
    threads against 140-165 when the pages are local (profiles/r05/cli_page_cache_node.txt).  The packed stream is a quarter of
    the bytes: better to read next to the pages and write across.  page_nodes: {node: sampled pages} (MmapChunker.page_nodes).
    Returns (node, set of CPUs) when most pages lie on one node other than the GPU's and this process may run there, else
    None."""
    if not _BOUND or not page_nodes:
        return None
    total = sum(page_nodes.values())
    node = max(page_nodes, key=page_nodes.get)
    if node < 0 or node == _BOUND["numa_node"] or page_nodes[node] < min_share * total:
        return None
    try:
        with open("%s/node%d/cpulist" % (sysfs, node)) as f:
            cpus = _parse_cpulist(f.read()) & _BOUND["before"]
    except (OSError, ValueError):
        return None
    return (node, cpus) if cpus else None
