#!/usr/bin/env python3
"""PCIe-inclusive rate: the batch lives in HOST memory (numpy) and every kmm_map_reads_uniform call
stages it to HBM (copy stream, double-buffered, overlapped with the previous call's kernel).
This is NOT bench.py's `value` (which starts with the reads resident in HBM); it goes into DESIGN.md."""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn          # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex         # noqa: E402


def main():
    n_index = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
    index, genome = syn.make_index(n_index, seed=1)
    bases, _ = syn.make_reads(genome, R, 150, seed=2)
    pinned = None
    try:
        import torch
        pinned = torch.from_numpy(bases).pin_memory()
    except Exception as exc:
        print("no pinned variant:", exc)
    with DeviceIndex.from_index(index) as dev:
        for name, buf in (("pageable numpy", bases), ("pinned (torch.pin_memory)", pinned)):
            if buf is None:
                continue
            dev.map_reads_uniform(buf, R, 150, 31)
            dev.synchronize()
            t0 = time.perf_counter()
            K = 5
            for _ in range(K):
                dev.map_reads_uniform(buf, R, 150, 31)
            dev.synchronize()
            dt = time.perf_counter() - t0
            print("%s: %d reads x %d calls in %.3f s -> %.1f M k-mers/s, %.2f GB/s of read bytes over PCIe"
                  % (name, R, K, dt, R * 120 * K / dt / 1e6, R * 150 * K / dt / 1e9))


if __name__ == "__main__":
    main()
