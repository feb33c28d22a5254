#!/usr/bin/env python3
"""bench.py — M k-mers mapped/sec on the fused reads -> node-counts hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (kmm_map_reads_uniform: encode -> rolling 31-mer pack ->
modulo -> bucket gather -> compare/filter -> atomic node counts) over one batch of synthetic reads
that is already resident in HBM.  Workload at N=1 (default, --config 2): BASELINE configs[2] — synthetic
150 bp reads in batches of 20 M (--steps batches: 5 by default = 100 M reads), k=31, 100 M-k-mer index (the
headline config; --config 1 = configs[1], the 10 M-k-mer index in 10 M-read batches; --index-kmers 1000000000 =
configs[4]'s index).
With N ranks every rank maps its own batch per step (reads shard by chunk: weak scaling) against a
replicated index and the per-rank uint32 count vectors are summed once with RCCL at the end of the job, inside the
timed region; the same invocation then also times BASELINE configs[3] — the SAME --steps x --reads reads split over
the ranks (`strong`) — and the PCIe-inclusive leg on every rank at once (`value_incl_h2d`).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per k-mer of the fused path at alpha=0.5, p=0.18, d=1:
# 150/120 read bytes + 8 bucket record + 8*(alpha + p*d) entry k-mers + 14*p*d (freq, node, count RMW)
B_ALG_PER_KMER = 1.25 + 8.0 + 8.0 * (0.5 + 0.18) + 14.0 * 0.18      # = 17.21
# The direct kernels do all of it in one launch.  The radix path does the same work as a
# pipeline of streaming kernels; its roofline line prices the WHOLE pipeline against the same 17.21 B / k-mer
# (sum of the kernels' average durations), and lists every kernel with the bytes it streams by design:
# pass 1 reads 1.25 B and writes 8 B per k-mer, pass 2 reads 8 and writes 8, pass 3 reads 8 (+ the index
# slices, once per batch), the directory scan and the flush are per-block / per-entry, not per k-mer.
B_ALG = {
    "k_map_reads": B_ALG_PER_KMER,
    "k_map_kmers": B_ALG_PER_KMER - 1.25 + 8.0,
}
RX_STREAM_BYTES = {"k_rx_p1": 1.25 + 8.0, "k_rx_p2": 16.0, "k_rx_p3": 8.0}
RX_KERNELS = ("k_rx_p1", "k_rx_scan", "k_rx_p2", "k_rx_p3")
HBM_PEAK_GBPS = 8000.0                                               # MI355X_MICROARCH.md


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class ClockSampler:
    """The GPU's engine / memory / fabric clocks as the driver reports them (the starred level of
    /sys/bus/pci/devices/<id>/pp_dpm_{sclk,mclk,fclk}), read every few milliseconds by a thread while the timed steps run:
    a kernel that takes 5.96 ms on one box and 6.47 on another is to be read beside the clocks it ran at."""

    def __init__(self, pci, period_s=0.004):
        import threading
        self.base = "/sys/bus/pci/devices/%s" % pci if pci else None
        self.period = period_s
        self.samples = {"sclk": [], "mclk": [], "fclk": []}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open("%s/pp_dpm_%s" % (self.base, name)) as f:
                for line in f:
                    if "*" in line:
                        return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            for name, got in self.samples.items():
                v = self._read(name)
                if v is not None:
                    got.append(v)
            self._stop.wait(self.period)

    def start(self):
        if self.base and os.path.exists(self.base + "/pp_dpm_sclk"):
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread.ident is not None:
            self._thread.join()

    def summary(self):
        out = {}
        for name, got in self.samples.items():
            if got:
                g = sorted(got)
                out[name] = {"min": g[0], "median": g[len(g) // 2], "max": g[-1], "samples": len(g)}
        return out or None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(1, 2),
                    help="BASELINE.json configs[]: 2 = 100 M-k-mer index (headline, default), 1 = 10 M-k-mer index")
    ap.add_argument("--reads", type=int, default=None,
                    help="reads per batch per GPU (default: 20 M at --config 2 — 5 steps = BASELINE configs[2]'s 100 M reads; "
                         "10 M at --config 1 = configs[1]'s chunk size; 10 M-read batches of configs[2]: --reads 10000000)")
    ap.add_argument("--index-kmers", type=int, default=None, help="overrides --config's index size")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--numpy-builder", action="store_true", help="build the synthetic index with numpy instead of kmm_build_index")
    ap.add_argument("--torch-index", action="store_true",
                    help="generate and build the synthetic index on the GPU (synthetic.make_index_torch; default above "
                         "3e8 k-mers: BASELINE configs[4]'s 1e9-k-mer index takes minutes on the host)")
    ap.add_argument("--modulo", type=int, default=None, help="hash-table size (default: smallest prime >= 2N)")
    ap.add_argument("-k", "--kmer-size", type=int, default=31)
    ap.add_argument("--skewed", action="store_true", help="node = i mod 1000 (atomic contention)")
    ap.add_argument("--path", type=int, default=0, help="0 auto (by batch size), 1 direct fused kernel, 2 radix path")
    ap.add_argument("--part-shift", type=int, default=None)
    ap.add_argument("--fine-bits", type=int, default=None, help="radix path: log2 fine partitions per coarse partition")
    ap.add_argument("--no-packed-tiles", action="store_true", help="radix pass 1: position-based tiles also for reads of one length")
    ap.add_argument("--no-radix-filter", action="store_true", help="radix pass 2 without the empty-bucket filter")
    ap.add_argument("--bucket-order-flush", action="store_true", help="radix path: flush the per-entry counts in bucket order (scattered atomics)")
    ap.add_argument("--grid-per-cu", type=int, default=None)
    ap.add_argument("--radix-grid-per-cu", type=int, default=None, help="persistent workgroups per CU of radix passes 2 and 3 (1 or 2)")
    ap.add_argument("--static-schedule", action="store_true", help="disable the dynamic tile queue")
    ap.add_argument("--dyn-chunk", type=int, default=None)
    ap.add_argument("--no-filter", action="store_true", help="disable the L2 occupancy-bitmap prefilter")
    ap.add_argument("--general-path", action="store_true", help="use kmm_map_reads with an offsets array")
    ap.add_argument("--records", action="store_true",
                    help="time kmm_map_records instead: each batch is a raw FASTQ chunk in HBM (<= 2^30 bytes), "
                         "records are parsed on the GPU")
    ap.add_argument("--operator", action="store_true",
                    help="time the operator path instead: k-mers extracted once (kmm_extract_kmers) into HBM, "
                         "each step = kmm_map_kmers over them (drop-in for map_kmers_to_graph_index)")
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000,
                    help="CPU baseline sample: a prefix of batch 0 (BASELINE.md section 3: 10 M-read prefix)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the PCIe-inclusive measurement (value_incl_h2d)")
    ap.add_argument("--host-pack-threads", type=int, default=None,
                    help="host threads that pack host-resident reads to 2 bits per base inside the map call (default: the library's "
                         "own default, min(16, CPU budget of the rank); 0 = the bytes cross PCIe as they are)")
    ap.add_argument("--no-records-host-leg", action="store_true",
                    help="skip the raw-FASTQ-in-host-memory measurement (config.records_from_host_memory)")
    ap.add_argument("--no-numa-bind", action="store_true", help="leave the process's CPU affinity alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the real multi-GPU path) or gloo (rehearsal on a 1-GPU box: all "
                         "ranks share device LOCAL_RANK %% device_count, the reduce runs on host copies)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (BASELINE configs[3]): the job is --steps x --reads reads IN TOTAL (default 5 x 20 M "
                         "= 100 M), split evenly over the ranks; timed region = map + flush + RCCL reduce as always. "
                         "Default (weak): every rank maps --steps batches of --reads reads.")
    ap.add_argument("--max-freq", type=int, default=1000,
                    help="max_index_lookup_frequency (-1 filters every hit: timing ablation without atomics)")
    args = ap.parse_args()
    if args.index_kmers is None:
        args.index_kmers = {1: 10_000_000, 2: 100_000_000}[args.config]
    if args.reads is None:
        # configs[1] names its chunk size (10 M reads); configs[2] is "100 M reads" on one GPU: batches sized for 288 GB of
        # HBM — every batch pays the radix passes' fixed costs once (152 / 165 / 164 G k-mers/s at
        # 10 / 20 / 30 M reads per batch)
        args.reads = 10_000_000 if args.config == 1 else 20_000_000

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev_t = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev_t)
            except TypeError:               # older torch without the device_id keyword
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.distributed import bind_to_gpu_numa_node, init_rccl_comm, reduce_node_counts
    from kmer_mapper_amd.engine import DeviceIndex

    # every rank next to its own GPU: the threads that pack its reads and the page-locked buffers they fill live on the
    # NUMA node the GPU hangs off (eight ranks pulling from host DRAM at once)
    if args.no_numa_bind:
        os.environ["KMM_NO_NUMA_BIND"] = "1"
    numa = bind_to_gpu_numa_node(local_rank)
    from kmer_mapper_amd import _io as kmm_io
    # this rank's share of the host's cores: the CPUs of its affinity mask split among the ranks bound to the same mask,
    # and its share of the cgroup's CPU quota where there is one
    ranks_here = 1
    if world > 1:
        masks = [None] * world
        dist.all_gather_object(masks, (numa["numa_node"] if numa["bound"] else -1))
        ranks_here = sum(1 for m in masks if m == masks[rank])
    quota = kmm_io.cpu_quota()
    cpu_budget = max(1, kmm_io.affinity_count() // ranks_here)
    if quota is not None:
        cpu_budget = max(1, min(cpu_budget, quota // world))

    k, L, R = args.kmer_size, args.read_len, args.reads
    t_setup = time.time()
    torch_index = args.torch_index or args.index_kmers > 300_000_000
    if torch_index:
        assert not args.skewed
        index, g_ascii = syn.make_index_torch(args.index_kmers, k=k, seed=1, modulo=args.modulo, device=local_rank)
    else:
        index, genome = syn.make_index(args.index_kmers, k=k, seed=1, skewed=args.skewed, modulo=args.modulo,
                                       gpu_builder=not args.numpy_builder, device=local_rank)   # untimed setup; identical arrays
    mx = index.max_node_id()
    n_entries, modulo_used = len(index._kmers), index._modulo
    log("index: %d entries, modulo %d, max_node_id %d (%.1fs)" % (n_entries, modulo_used, mx, time.time() - t_setup))
    dev = DeviceIndex.from_index(index, mx, device=local_rank)
    if torch_index:          # the handle owns its own copy: keep only what the CPU leg needs (host arrays, on demand)
        index = index.to_host() if (rank == 0 and not args.no_cpu_baseline) else None
        torch.cuda.empty_cache()
    dev.set_param("path", args.path)
    # the host cores' share (kmm_set_param "host_pack_threads"): the library's default is min(16, budget of the process);
    # with several ranks on one host every rank takes its share of the cores
    if args.host_pack_threads is not None:
        dev.set_param("host_pack_threads", args.host_pack_threads)
    elif world > 1:
        dev.set_param("host_pack_threads", min(16, cpu_budget) if cpu_budget >= 4 else 0)
    n_pack_default = dev.get_param("host_pack_threads")
    if args.no_filter:
        dev.set_param("occupancy_filter", 0)
    if args.dyn_chunk is not None:
        dev.set_param("dyn_chunk", args.dyn_chunk)
    if args.static_schedule:
        dev.set_param("dynamic_schedule", 0)
    if args.grid_per_cu is not None:
        dev.set_param("grid_per_cu", args.grid_per_cu)
    if args.part_shift is not None:
        dev.set_param("part_shift", args.part_shift)
    if args.no_radix_filter:
        dev.set_param("radix_filter", 0)
    if args.no_packed_tiles:
        dev.set_param("radix_packed_tiles", 0)
    if args.fine_bits is not None:
        dev.set_param("fine_bits", args.fine_bits)
    if args.radix_grid_per_cu is not None:
        dev.set_param("radix_grid_per_cu", args.radix_grid_per_cu)
    if args.bucket_order_flush:
        dev.set_param("radix_sorted_flush", 0)
    counts = torch.zeros(mx + 1, dtype=torch.int32, device=dev_t)    # uint32 bits; wrap-add == int32 add
    dev.bind_counts(counts)

    if not torch_index:
        g_ascii = torch.from_numpy(syn.ACGT[genome]).to(dev_t)
    batches = [syn.make_reads_torch(g_ascii, R, L, seed=1000 * (rank + 1) + b) for b in range(2)]
    offs = None
    if args.general_path:
        offs = torch.arange(R + 1, dtype=torch.int64, device=dev_t) * L
    del g_ascii
    torch.cuda.synchronize()
    kmers_per_step = R * max(L - k + 1, 0)
    fastq_batches = None
    if args.records:
        from kmer_mapper_amd import _lib as kmm_lib
        rec_len = 4 + L + 3 + L + 1                                   # "@rd\n" seq "\n+\n" qual "\n"
        # (kmm_map_records takes buffers of any size: it works through them in pieces of at most 2^30 bytes cut at
        # record boundaries)
        fastq_batches = []
        for b in batches:
            rec = torch.empty((R, rec_len), dtype=torch.uint8, device=dev_t)
            rec[:, 0:4] = torch.tensor(list(b"@rd\n"), dtype=torch.uint8, device=dev_t)
            rec[:, 4:4 + L] = b.view(R, L)
            rec[:, 4 + L:4 + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev_t)
            rec[:, 4 + L + 3:4 + L + 3 + L] = ord("F")
            rec[:, -1] = 10
            fastq_batches.append(rec.reshape(-1))
        torch.cuda.synchronize()
    kmer_batches = None
    if args.operator:
        from kmer_mapper_amd.engine import extract_kmers
        o = torch.arange(R + 1, dtype=torch.int64, device=dev_t) * L
        kmer_batches = []
        for b in batches:
            km = torch.empty(kmers_per_step, dtype=torch.int64, device=dev_t)     # uint64 bit patterns
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            extract_kmers(b, o, k, device=local_rank, out=km)
            log("kmm_extract_kmers: %.1f ms for %d k-mers" % ((time.perf_counter() - t_e) * 1e3, kmers_per_step))
            kmer_batches.append(km)
        del o
    log("setup done in %.1fs; %d reads/batch/GPU, %d k-mers/step/GPU" % (time.time() - t_setup, R, kmers_per_step))

    # batch sizes of this rank: weak scaling = --steps full batches; strong scaling = this rank's share of the
    # --steps x --reads reads of the whole job, cut into batches of at most --reads reads
    if args.strong:
        from kmer_mapper_amd.distributed import shard_range
        lo_r, hi_r = shard_range(R * args.steps, rank, world)
        sizes = [R] * ((hi_r - lo_r) // R) + ([(hi_r - lo_r) % R] if (hi_r - lo_r) % R else [])
        assert not (args.records or args.operator), "--strong times the fused reads path"
    else:
        sizes = [R] * args.steps
    my_kmers = sum(sizes) * max(L - k + 1, 0)

    def step(i):
        b = batches[i & 1]
        if sizes[i] != R:
            if offs is not None:
                dev.map_reads(b[:sizes[i] * L], offs[:sizes[i] + 1], k, args.max_freq)
            else:
                dev.map_reads_uniform(b[:sizes[i] * L], sizes[i], L, k, args.max_freq)
            return
        if fastq_batches is not None:
            used, n_rec = dev.map_records(fastq_batches[i & 1], fmt=kmm_lib.FORMAT_FASTQ, k=k,
                                          max_index_lookup_frequency=args.max_freq)
            assert n_rec == R
        elif kmer_batches is not None:
            dev.map_kmers(kmer_batches[i & 1], args.max_freq)
        elif offs is not None:
            dev.map_reads(b, offs, k, args.max_freq)
        else:
            dev.map_reads_uniform(b, R, L, k, args.max_freq)

    def fence():
        dev.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    own_comm = False
    comm_note = ""
    if world > 1 and args.dist_backend == "nccl":
        # the library's own communicator (kmm_comm_init_rank); every rank must have it, else all of them reduce
        # through torch.distributed (same RCCL sum of the same vector)
        try:
            init_rccl_comm(dev)
            ok = 1
        except Exception as e:                       # noqa: BLE001 - any failure means "fall back", on every rank
            print("rank %d: kmm_comm_init_rank failed (%s): reducing through torch.distributed" % (rank, e), file=sys.stderr, flush=True)
            comm_note = "rank %d: %s" % (rank, str(e)[:200])
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev_t)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        own_comm = bool(flag.item())
        if not own_comm:                             # which rank(s) could not join, and why: the record says which reduce ran
            notes = [None] * world
            dist.all_gather_object(notes, comm_note)
            comm_note = "; ".join(n for n in notes if n) or "a rank failed to join"
    elif world > 1:
        comm_note = "--dist-backend %s (rehearsal: the sum runs on host copies)" % args.dist_backend

    def reduce_counts(t):
        if args.dist_backend == "nccl":
            # RCCL sum of the uint32 count vectors over xGMI, behind the C ABI (kmm_comm_reduce_counts): in place
            # on the bound vector `counts`
            if t is counts and own_comm:
                dev.comm_reduce_counts(root=0)
            else:
                reduce_node_counts(t, dst=0)
        else:                                        # rehearsal only
            h = t.cpu()
            reduce_node_counts(h, dst=0)
            t.copy_(h)

    for i in range(args.warmup):
        step(i)
    if world > 1:                                   # warm the communicator the timed reduce will use (the first collective
        reduce_counts(counts)                       # of an RCCL communicator sets up its channels); counts are zeroed below
    fence()
    counts.zero_()
    fence()

    dev.set_timing(True)
    clocks = ClockSampler(numa.get("pci"))           # (the clocks beside every figure, box to box)
    clocks.start()
    t0 = time.perf_counter()
    for i in range(len(sizes)):
        step(i)
    dev.synchronize()
    t_map = time.perf_counter()
    clocks.stop()
    if world > 1:
        reduce_counts(counts)
    fence()
    t1 = time.perf_counter()
    dev.set_timing(False)
    timing = dev.get_timing()

    elapsed = t1 - t0
    reduce_s = t1 - t_map
    per_rank_map_ms = [round((t_map - t0) * 1e3, 3)]
    if world > 1:
        cdev = dev_t if args.dist_backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed, reduce_s], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, reduce_s = tt[0].item(), tt[1].item()
        mine = torch.tensor([(t_map - t0) * 1e3, float(my_kmers)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_map_ms = [round(t[0].item(), 3) for t in every]
        total_kmers = int(sum(t[1].item() for t in every))
    else:
        total_kmers = my_kmers
    # SURVEY 8(d) counts "final D2H/reduce" into the map phase; the reference's own GPU timer stops before it
    # (command_line_interface.py:78-79), the bench contract's `value` likewise: reported beside it
    final_d2h_ms = 0.0
    if rank == 0:                      # into page-locked memory the library hands out (kmm_host_alloc): the link's rate
        from kmer_mapper_amd import _lib as _kl
        import numpy as _np
        landing = _kl.pinned_array(mx + 1, _np.uint32)
        final_d2h_ms = None
        for _ in range(2):
            td0 = time.perf_counter()
            dev.get_node_counts(out=landing)
            t_ = (time.perf_counter() - td0) * 1e3
            final_d2h_ms = t_ if final_d2h_ms is None else min(final_d2h_ms, t_)
        del landing
    # ... and into an ORDINARY numpy array, what the reference's caller gets: kmm_get_node_counts takes a large vector through
    # the handle's page-locked staging ring (the first call makes the ring and takes the array's first-touch page faults)
    final_d2h_pageable_ms = None
    if rank == 0:
        import numpy as _np
        landing = _np.empty(mx + 1, dtype=_np.uint32)
        for _ in range(2):
            td2 = time.perf_counter()
            dev.get_node_counts(out=landing)
            t_ = (time.perf_counter() - td2) * 1e3
            final_d2h_pageable_ms = t_ if final_d2h_pageable_ms is None else min(final_d2h_pageable_ms, t_)
        del landing
    hits = int(counts.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) if rank == 0 else 0
    hbm_free, hbm_total = torch.cuda.mem_get_info(dev_t)
    survivors = None
    if dev.get_param("radix_filter"):
        p2_in, p2_drop = dev.get_param("radix_p2_kmers"), dev.get_param("radix_p2_dropped")
        if p2_in:
            survivors = 1.0 - p2_drop / p2_in

    result = None
    if rank == 0:
        value = total_kmers / elapsed / 1e6
        # dominant kernel = the hot-path kernel with the largest summed HIP-event time; the radix path is a
        # pipeline of streaming kernels (one launch each per step) and is priced as a whole
        dom = max(timing, key=lambda n: timing[n][0])
        per_kernel = None
        if dom.startswith("k_rx"):
            used = [n for n in RX_KERNELS if timing[n][1]]
            launches = timing["k_rx_p1"][1]                      # one pipeline (one launch of every kernel) per step
            avg_kernel_s = sum(timing[n][0] for n in used) / max(launches, 1) / 1e3
            kmers_per_launch = my_kmers / max(launches, 1)
            per_kernel = {}
            # with the empty-bucket filter pass 2 writes, and pass 3 reads, only the surviving fraction of the k-mers
            stream_bytes = dict(RX_STREAM_BYTES)
            if survivors is not None:
                stream_bytes["k_rx_p2"] = 8.0 + 8.0 * survivors
                stream_bytes["k_rx_p3"] = 8.0 * survivors
            for n in used:
                t_s = timing[n][0] / max(launches, 1) / 1e3
                per_kernel[n] = {"avg_ms": round(t_s * 1e3, 3)}
                if n in stream_bytes:
                    gbps = kmers_per_launch * stream_bytes[n] / t_s / 1e9
                    per_kernel[n].update(stream_bytes_per_kmer=round(stream_bytes[n], 2), stream_GB_per_s=round(gbps, 1),
                                         frac_of_hbm_peak=round(gbps / HBM_PEAK_GBPS, 4))
            dom = "+".join(used)
            b_alg = B_ALG_PER_KMER if not args.operator else B_ALG["k_map_kmers"]
        else:
            kernel_ms, launches = timing[dom]
            avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
            kmers_per_launch = my_kmers / max(launches, 1)
            b_alg = B_ALG[dom]
        achieved = kmers_per_launch * b_alg / avg_kernel_s / 1e9
        traffic = None
        sector = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and not args.skewed:
            try:
                tj_all = json.load(open(tpath))
                entries = tj_all.get("entries", [tj_all]) if isinstance(tj_all, dict) else tj_all
                for tj in entries:      # one committed PMC pass per (kernel, batch size, index size)
                    if (tj.get("reads") != R or tj.get("index_kmers") != args.index_kmers
                            or tj.get("kernel") != dom):
                        continue
                    traffic = tj.get("hbm_bytes_per_launch")
                    if per_kernel and "per_kernel" in tj:          # measured HBM bytes of every kernel of the pipeline
                        for n, pk in per_kernel.items():
                            m = [v for kk, v in tj["per_kernel"].items()
                                 if kk == n or kk == n + "f" or (n == "k_rx_scan" and kk in ("k_rx_colsum", "k_rx_chunkscan", "k_rx_tables",
                                                                            "k_rx_colscan", "k_rx_tr2"))]
                            if m:
                                b = sum(v["read_bytes"] + v["write_bytes"] for v in m)
                                pk["measured_hbm_bytes"] = round(b)
                                pk["measured_hbm_GB_per_s"] = round(b / (pk["avg_ms"] * 1e-3) / 1e9, 1)
                    if "l2_miss_read_requests_per_kmer" in tj:
                        # request-granular view: random probes are bound by the NUMBER of
                        # requests — ~55 G/s for L2-missing reads, ~254 G/s for L2 hits, and the two add
                        miss, hit = tj["l2_miss_read_requests_per_kmer"], tj["l2_hits_per_kmer"]
                        sector = {
                            "l2_miss_read_requests_per_kmer": round(miss, 3),
                            "l2_hits_per_kmer": round(hit, 3),
                            "miss_ceiling_Greq_per_s": 55.0, "hit_ceiling_Greq_per_s": 254.0,
                            "model_ms_per_launch": round(kmers_per_launch * (miss / 55e9 + hit / 254e9) * 1e3, 2),
                            "hbm_bytes_per_kmer": round(traffic / kmers_per_launch, 1),
                            "measured_hbm_GB_per_s": round(traffic / avg_kernel_s / 1e9, 1),
                        }
                    break
            except Exception:
                traffic = None
        if args.index_kmers == 1_000_000_000:
            cfg_label = "configs[4]'s index on one GPU (1 B-k-mer index resident in HBM)"
        elif args.index_kmers == 100_000_000:
            cfg_label = "configs[2] (100 M-k-mer index; %d reads = %d batches of %d)" % (R * args.steps, args.steps, R)
        elif args.index_kmers == 10_000_000 and R == 10_000_000:
            cfg_label = "configs[1] (10 M reads per batch, 10 M-k-mer index)"
        else:
            cfg_label = "custom"
        result = {
            "metric": "M k-mers mapped/sec (whole node), k=%d %dbp reads" % (k, L),
            "value": round(value, 1),
            "unit": "M k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d steps x %d synthetic %d bp reads per batch per GPU, k=%d, %d-k-mer index "
                            "(modulo %d, %d entries), reads resident in HBM, %s"
                            % (cfg_label, args.steps, R, L, k, args.index_kmers, modulo_used, n_entries,
                               "kmm_map_records on raw FASTQ chunks" if args.records else
                               "operator kmm_map_kmers on pre-extracted k-mers" if args.operator else
                               "fused kmm_map_reads" + ("" if args.general_path else "_uniform")),
                "kmers_per_step_per_gpu": kmers_per_step,
                "reads_in_hbm_when_timed": True, "h2d_in_timed_region": False,
                "job": ("strong scaling: %d reads in total split over %d rank(s), batches of <= %d reads"
                        % (R * args.steps, world, R)) if args.strong else
                       ("weak scaling: every rank maps %d batches of %d reads" % (args.steps, R)),
                "batches_this_rank": len(sizes),
                "per_rank_map_ms": per_rank_map_ms,
                "flush_ms_total": round(timing["k_rx_flush"][0], 3),
                "hit_rate": round(hits / max(total_kmers, 1), 4),
                "nodes": "skewed(mod 1000)" if args.skewed else "uniform",
                "path": {0: "auto", 1: "direct", 2: "radix"}[args.path],
                "path_taken": "radix" if per_kernel else "direct",
                "hbm_in_use_GB": round((hbm_total - hbm_free) / 1e9, 1),
                "index_views": {"radix_GB": round(dev.get_param("radix_view_bytes") / 1e9, 2),
                                "direct_GB": round(dev.get_param("direct_view_bytes") / 1e9, 2),
                                "direct_resident": bool(dev.get_param("direct_view_resident"))},
                "radix_filter": bool(dev.get_param("radix_filter")),
                "radix_filter_survivors": None if survivors is None else round(survivors, 4),
                "radix_packed_tiles": bool(dev.get_param("radix_packed_tiles")),
                "radix_fine_partitions": dev.get_param("n_partitions"),
                "radix_coarse_partitions": dev.get_param("n_coarse_partitions"),
                "occupancy_filter": bool(dev.get_param("occupancy_filter")),
                "occupancy_bits_per_bucket": dev.get_param("occupancy_bits_per_bucket"),
                "bloom_filter_bytes": dev.get_param("bloom_filter_bytes"),
                "final_reduce_ms": round(reduce_s * 1e3, 3) if world > 1 else 0.0,
                "final_d2h_ms": round(final_d2h_ms, 3),
                "final_d2h_pageable_ms": None if final_d2h_pageable_ms is None else round(final_d2h_pageable_ms, 3),
                "final_d2h_note": "%d-byte count vector -> page-locked host memory, outside the timed region (the reference's own "
                                  "timer stops before the fetch, command_line_interface.py:78-79)" % (4 * (mx + 1)),
                "parallelism": "reads sharded by batch over %d GPU(s), index replicated, one RCCL sum (%s)"
                               % (world, "kmm_comm_reduce_counts" if own_comm or world == 1 else
                                  "torch.distributed reduce, because kmm_comm_init_rank did not come up on every rank: " + comm_note),
                "numa_binding": numa,
                "gpu_clocks_MHz_during_timed_region": clocks.summary(),
                "host_cores_of_this_rank": cpu_budget,
                "kernel_ms_per_step": {n: round(t[0] / max(len(sizes), 1), 3) for n, t in timing.items() if t[1]},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_kmer": round(b_alg, 2),
                "avg_kernel_ms": round(avg_kernel_s * 1e3, 3),
                "launches": launches,
                "kernel_gkmers_per_s": round(kmers_per_launch / avg_kernel_s / 1e9, 2),
                "request_model": sector,
                "per_kernel": per_kernel,
            },
        }

    # ---- N > 1: the same invocation also answers BASELINE configs[3] and the PCIe question ------------------------
    # (i) `strong`: the SAME --steps x --reads reads as ONE job split evenly over the ranks (reference
    #     command_line_interface.py:110-130: a fixed job cut over the workers, one additive reduce), timed region = map +
    #     flush + RCCL reduce; efficiency against the N = 1 time the weak leg measured on every rank (each rank mapped the
    #     whole --steps x --reads job there).
    # (ii) the staged leg on EVERY rank at once: batches in pinned host memory, copied to HBM by every call over the
    #     rank's own PCIe link (all ranks pull from host DRAM together: 8 x ~55 GB/s).
    def timed_job(job_sizes, get_batch):
        counts.zero_()
        fence()
        dev.set_timing(True)
        a0 = time.perf_counter()
        for j, n_j in enumerate(job_sizes):
            dev.map_reads_uniform(get_batch(j)[: n_j * L], n_j, L, k, args.max_freq)
        dev.synchronize()
        a_map = time.perf_counter()
        if world > 1:
            reduce_counts(counts)
        fence()
        a1 = time.perf_counter()
        dev.set_timing(False)
        tm = dev.get_timing()
        n_k = sum(job_sizes) * max(L - k + 1, 0)
        cdev = dev_t if (world > 1 and args.dist_backend == "nccl") else "cpu"
        mine = torch.tensor([(a_map - a0) * 1e3, tm["k_rx_flush"][0], (a1 - a_map) * 1e3, (a1 - a0) * 1e3, float(n_k),
                             float(sum(job_sizes))], dtype=torch.float64, device=cdev)
        every = [mine]
        if world > 1:
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
        cols = [[round(t[c].item(), 3) for t in every] for c in range(6)]
        return {"map_ms": cols[0], "flush_ms": cols[1], "reduce_ms": cols[2], "elapsed_ms": cols[3],
                "kmers": int(sum(cols[4])), "reads": [int(x) for x in cols[5]]}

    plain_fused = not (args.records or args.operator or args.general_path)
    if world > 1 and plain_fused and not args.strong:
        from kmer_mapper_amd.distributed import shard_range
        lo_r, hi_r = shard_range(R * args.steps, rank, world)
        n_b = max(1, -(-(hi_r - lo_r) // R))                      # equal batches of at most R reads (no small last one)
        s_sizes = [(hi_r - lo_r) // n_b + (1 if j < (hi_r - lo_r) % n_b else 0) for j in range(n_b)]
        st = timed_job(s_sizes, lambda j: batches[j & 1])
        if rank == 0:
            t_strong = max(st["elapsed_ms"])
            t_n1 = sum(per_rank_map_ms) / len(per_rank_map_ms)        # weak leg: map + flush of the whole job on ONE GPU
            result["strong"] = {
                "workload": "BASELINE configs[3]: %d reads in total split over %d ranks (%s reads per rank), batches of <= %d"
                            % (R * args.steps, world, "/".join(str(x) for x in sorted(set(st["reads"]))), R),
                "value": round(st["kmers"] / (t_strong * 1e-3) / 1e6, 1), "unit": "M k-mers/s",
                "elapsed_ms": round(t_strong, 3),
                "per_rank_map_ms": st["map_ms"],                      # map kernels + the flush behind the last batch
                "flush_ms": st["flush_ms"], "final_reduce_ms": round(max(st["reduce_ms"]), 3),
                "n1_time_ms_measured_in_the_weak_leg": round(t_n1, 3),
                "efficiency_vs_n1": round(t_n1 / (world * t_strong), 4),
                "timed_region": "map + flush + RCCL reduce, barrier / synchronize on both sides, max over ranks",
            }
    # ---- SURVEY 8(d)'s map-phase figure: the SAME steps with the reads in (page-locked) HOST memory, handed to the map call
    # as the reference hands a chunk to a worker (command_line_interface.py:109-111).  `value` above times reads that are
    # already resident in HBM (the bench contract).  By default the call packs the reads to 2 bits per base on the rank's
    # host threads first ("host_pack_threads": the reference's -t, its default 16 where the rank has the cores) and a quarter
    # of the bytes crosses PCIe: `value_incl_h2d`; with 0 threads the ASCII bytes cross as they are (~55 GB/s per link):
    # `value_incl_h2d_plain_copy`.  At N > 1 every rank runs the leg at once, each over its own link, reduce included.
    if plain_fused and not args.no_h2d_leg:
        host_batches = [b.cpu().pin_memory() for b in batches]

        def staged_leg(threads):
            dev.set_param("host_pack_threads", threads)
            dev.reset()
            dev.map_reads_uniform(host_batches[0], R, L, k, args.max_freq)      # warm the staging / page-locked buffers
            dev.synchronize()
            packed_before = dev.get_param("host_packed_calls")
            job = timed_job(sizes, lambda j: host_batches[j & 1])
            job["packed_calls"] = dev.get_param("host_packed_calls") - packed_before
            return job

        legs = {"plain": staged_leg(0)}
        if n_pack_default > 0:
            legs["packed"] = staged_leg(n_pack_default)
        dev.set_param("host_pack_threads", n_pack_default)
        del host_batches
        if rank == 0:
            def describe(job, what):
                t = max(job["elapsed_ms"])
                return round(job["kmers"] / (t * 1e-3) / 1e6, 1), {
                    "what": what,
                    "per_rank_read_GB_per_s": [round(r_ * L / (m_ * 1e-3) / 1e9, 1) for r_, m_ in zip(job["reads"], job["map_ms"])],
                    "host_read_GB_per_s_total": round(sum(job["reads"]) * L / (t * 1e-3) / 1e9, 1),
                    "elapsed_ms": round(t, 3)}
            v_plain, d_plain = describe(legs["plain"], "every batch in page-locked host memory, copied to HBM as ASCII bytes by the call "
                                        "that maps it (double-buffered staging on a copy stream)%s"
                                        % (", on all %d ranks at once; reduce included" % world if world > 1 else ""))
            result["value_incl_h2d_plain_copy"] = v_plain
            result["config"]["h2d_leg_plain_copy"] = d_plain
            v_def, d_def, frac_key = v_plain, d_plain, "plain copy (no host threads to pack with)"
            if "packed" in legs and legs["packed"]["packed_calls"] == len(sizes):
                v_def, d_def = describe(legs["packed"], "every batch in host memory, packed to 2 bits per base by %d host threads per rank "
                                        "inside the call (kmm_set_param host_pack_threads; library default = min(16, the rank's cores): "
                                        "this rank has %d), a quarter of the bytes over PCIe%s"
                                        % (n_pack_default, cpu_budget, ", on all %d ranks at once; reduce included" % world if world > 1 else ""))
                d_def["host_pack_threads"] = n_pack_default
                d_def["pcie_GB_per_s_total"] = round(d_def["host_read_GB_per_s_total"] / 4, 1)
                frac_key = "reads packed to 2 bits per base by %d host threads per rank" % n_pack_default
                result["value_incl_h2d_host_packed"] = v_def
            result["value_incl_h2d"] = v_def
            result["config"]["value_incl_h2d"] = v_def
            result["config"]["h2d_leg"] = d_def
            result["config"]["h2d_in_value_incl_h2d"] = frac_key
            result["roofline"]["frac_incl_h2d"] = round(v_def * 1e6 * B_ALG_PER_KMER / 1e9 / (HBM_PEAK_GBPS * world), 4)
            result["roofline"]["frac_incl_h2d_plain_copy"] = round(v_plain * 1e6 * B_ALG_PER_KMER / 1e9 / (HBM_PEAK_GBPS * world), 4)
            result["roofline"]["note"] = ("frac prices the HBM-resident pipeline (sum of the kernels' average durations); "
                                          "frac_incl_h2d the whole map phase with the reads starting in host memory (wall clock)")

    # ---- raw FASTQ in host memory -> kmm_map_records (what `kmer_mapper map` does with a file mapping): the host threads pack
    # the sequence lines, the GPU maps; N = 1 only, 10 M-record chunks (3.08 GB each)
    if rank == 0 and world == 1 and plain_fused and not args.no_records_host_leg and n_pack_default > 0:
        from kmer_mapper_amd import _lib as kmm_lib
        R_rec = min(R, 10_000_000)
        rec_len = 4 + L + 3 + L + 1
        host_fq = []
        for b in batches:
            rec = torch.empty((R_rec, rec_len), dtype=torch.uint8, device=dev_t)
            rec[:, 0:4] = torch.tensor(list(b"@rd\n"), dtype=torch.uint8, device=dev_t)
            rec[:, 4:4 + L] = b[:R_rec * L].view(R_rec, L)
            rec[:, 4 + L:4 + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev_t)
            rec[:, 4 + L + 3:4 + L + 3 + L] = ord("F")
            rec[:, -1] = 10
            host_fq.append(rec.reshape(-1).cpu().numpy())         # pageable host memory, like a file mapping
            del rec
        n_steps_rec = max(2, min(len(sizes) * R // R_rec, 8))
        dev.reset()
        dev.map_records(host_fq[0], fmt=kmm_lib.FORMAT_FASTQ, k=k, max_index_lookup_frequency=args.max_freq)
        dev.synchronize()
        before = dev.get_param("host_packed_record_calls")
        tr0 = time.perf_counter()
        for i in range(n_steps_rec):
            used, n_rec = dev.map_records(host_fq[i & 1], fmt=kmm_lib.FORMAT_FASTQ, k=k, max_index_lookup_frequency=args.max_freq)
            assert n_rec == R_rec
        dev.synchronize()
        tr = time.perf_counter() - tr0
        result["config"]["records_from_host_memory"] = {
            "what": "%d calls of kmm_map_records on %d-record raw FASTQ chunks (%.2f GB each) in pageable host memory: the sequence "
                    "lines are packed to 2 bits per base by %d host threads inside the call (kmm_hostpack.hpp RecordsJob), the "
                    "stream crosses PCIe, the radix path maps it" % (n_steps_rec, R_rec, host_fq[0].nbytes / 1e9, n_pack_default),
            "M_kmers_per_s": round(n_steps_rec * R_rec * max(L - k + 1, 0) / tr / 1e6, 1),
            "fastq_GB_per_s": round(n_steps_rec * host_fq[0].nbytes / tr / 1e9, 1),
            "host_packed_calls": dev.get_param("host_packed_record_calls") - before}
        del host_fq

    # ---- CPU baseline + parity on a bounded sample (rank 0, N=1 only) ---------------------------
    if rank == 0 and not args.no_cpu_baseline and index is not None:
        from oracle import oracle        # checker / reported baseline only
        oracle_flags = oracle.use_native_build()      # -O3 -march=native, built on the box that runs it (setup.py:10-15)
        n_s = min(args.cpu_sample_reads, R)
        sample = batches[0][: n_s * L].cpu().numpy()
        s_offs = np.arange(n_s + 1, dtype=np.int64) * L
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        # reference CLI default -t 16; fewer when 16 private count vectors (4 B x nodes each) would not fit 24 GB
        n_threads = max(1, min(16, avail, int(24e9 // (4 * (mx + 1)))))
        oracle.map_reads(index, mx, sample[: 20000 * L], s_offs[:20001], k, n_threads=n_threads)  # warm
        tc0 = time.perf_counter()
        expect, n_k = oracle.map_reads(index, mx, sample, s_offs, k, n_threads=n_threads)
        tc = time.perf_counter() - tc0
        counts.zero_()
        torch.cuda.synchronize()
        dev.map_reads_uniform(batches[0][: n_s * L], n_s, L, k)
        dev.synchronize()
        got = counts.cpu().numpy().view(np.uint32)
        parity = bool(np.array_equal(got, expect))
        result["cpu_baseline"] = {
            "value": round(n_k / tc / 1e6, 2),
            "unit": "M k-mers/s",
            "cores": n_threads,
            "kind": "port",
            "sample": "first %d reads of batch 0 (%d k-mers, %.1f s wall); oracle/kmm_oracle.c "
                      "oracle_map_reads, gcc %s, %d pthreads (the reference CLI's -t default; more threads are SLOWER with "
                      "one private %d-byte count vector each, see all_cores), private count vectors summed"
                      % (n_s, n_k, tc, oracle_flags, n_threads, 4 * (mx + 1)),
        }
        # BASELINE.md section 3 also promises "all physical cores": every core this process may run on,
        # capped so that the private count vectors (4 B x nodes per thread) stay within 32 GB
        n_all = min(avail, max(1, int(32e9 // (4 * (mx + 1)))))
        if n_all > n_threads and world == 1:          # (N > 1: the other ranks wait at the barrier; the 16-thread leg is the baseline)
            ta0 = time.perf_counter()
            expect_all, _ = oracle.map_reads(index, mx, sample, s_offs, k, n_threads=n_all)
            ta = time.perf_counter() - ta0
            result["cpu_baseline"]["all_cores"] = {
                "value": round(n_k / ta / 1e6, 2), "cores": n_all, "wall_s": round(ta, 1),
                "note": "same structure as the reference's pool (one private count vector per worker, summed): with "
                        "%d-byte vectors more workers are not faster; `best` is the faster of the two legs"
                        % (4 * (mx + 1))}
            result["cpu_baseline"]["best"] = max(result["cpu_baseline"]["value"], round(n_k / ta / 1e6, 2))
            parity = parity and bool(np.array_equal(expect_all, expect))
        result["cpu_baseline"]["available_cores"] = avail
        result["speedup_vs_cpu_16_threads"] = round(result["value"] / max(result["cpu_baseline"]["value"], 1e-9), 1)
        if "value_incl_h2d" in result:
            result["speedup_vs_cpu_16_threads_incl_h2d"] = round(result["value_incl_h2d"] / max(result["cpu_baseline"]["value"], 1e-9), 1)
        result["parity_vs_oracle_on_sample"] = parity
        if not parity:
            log("PARITY FAILURE: GPU counts differ from the oracle on the CPU sample")
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result.get("parity_vs_oracle_on_sample") is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
