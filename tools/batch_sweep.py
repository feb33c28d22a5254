#!/usr/bin/env python3
"""Throughput of back-to-back kmm_map_reads_uniform calls as a function of the batch size
(reads resident in HBM), to show where the launch/latency-bound regime ends."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn          # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex         # noqa: E402


def main():
    index, genome = syn.make_index(10_000_000, seed=1)
    g = torch.from_numpy(syn.ACGT[genome]).cuda()
    big = syn.make_reads_torch(g, 10_000_000, 150, seed=7)
    torch.cuda.synchronize()
    with DeviceIndex.from_index(index) as dev:
        print("%12s %10s %12s %14s" % ("reads/call", "calls", "ms/call", "G k-mers/s"))
        for R in (1_000, 8_000, 64_000, 500_000, 2_000_000, 10_000_000):
            calls = max(3, min(2000, 40_000_000 // R))
            view = big[: R * 150]
            for _ in range(3):
                dev.map_reads_uniform(view, R, 150, 31)
            dev.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                dev.map_reads_uniform(view, R, 150, 31)
            dev.synchronize()
            dt = time.perf_counter() - t0
            print("%12d %10d %12.4f %14.2f" % (R, calls, dt / calls * 1e3, R * 120 * calls / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
