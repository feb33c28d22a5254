#!/usr/bin/env python3
"""Soak of the host-packed staging route ("host_pack_threads"): random slices of a host-resident batch, random thread
counts, uniform and ragged calls, every call's counts compared with the direct kernel's on the same reads.
    python tools/hostpack_soak.py [rounds=60] [n_index=10000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_index = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    L, k = 150, 31
    index, genome = syn.make_index(n_index, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    R = 3_000_000
    bases, _ = syn.make_reads(genome, R, L, seed=4242)
    rng = np.random.default_rng(99)
    fails = 0
    t0 = time.perf_counter()
    with DeviceIndex.from_index(index, mx) as dev, DeviceIndex.from_index(index, mx) as ref:
        ref.set_param("path", 1)
        min_reads = dev.get_param("radix_min_units") // L + 1
        for r in range(rounds):
            n = int(rng.integers(min_reads, R + 1))
            first = int(rng.integers(0, R - n + 1))
            nt = int(rng.integers(1, 17))
            view = bases[first * L:(first + n) * L]
            dev.set_param("host_pack_threads", nt)
            slot_kb = int(rng.choice([0, 0, 16, 64, 1024, 4096]))   # the staging ring's slots (0: 16 MiB), wrapped up to hundreds of times
            dev.set_param("debug_ring_slot_kb", slot_kb)
            dev.reset()
            ref.reset()
            before = dev.get_param("host_packed_calls")
            if r % 3 == 2:          # ragged: the same bytes cut into reads of varying length
                cuts = np.sort(rng.choice(np.arange(1, view.shape[0]), size=n // 2, replace=False))
                offs = np.concatenate([[0], cuts, [view.shape[0]]]).astype(np.int64)
                dev.map_reads(view, offs, k)
                ref.map_reads(view, offs, k)
            else:
                dev.map_reads_uniform(view, n, L, k)
                ref.map_reads_uniform(view, n, L, k)
            got, want = dev.get_node_counts(), ref.get_node_counts()
            ok = np.array_equal(got, want) and dev.get_param("host_packed_calls") == before + 1
            fails += 0 if ok else 1
            print("round %d: %d reads from %d, %d threads, ring slots of %d KiB, %s: %s" % (r, n, first, nt, slot_kb or 16384, "ragged" if r % 3 == 2 else "uniform",
                                                                      "same as the direct kernel" if ok else "DIFFERENT"), flush=True)
    print("failures: %d of %d (%.1f s)" % (fails, rounds, time.perf_counter() - t0))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
