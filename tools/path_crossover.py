#!/usr/bin/env python3
"""Where does the radix path overtake the direct kernel?  Back-to-back kmm_map_reads_uniform calls of R reads
(resident in HBM) on path 1 and path 2 for a range of R; the auto rule (radix_min_units) should sit at the crossover.
    python tools/path_crossover.py [n_index] [reads_per_call path]
With the two extra arguments only that size runs on that path (1 direct, 2 radix): the form to put under
`rocprofv3 --kernel-trace --stats` for the radix path's fixed cost per call, kernel by kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn          # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex         # noqa: E402


def main():
    n_index = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    index, genome = syn.make_index(n_index, seed=1, gpu_builder=True)
    g = torch.from_numpy(syn.ACGT[genome]).cuda()
    big = syn.make_reads_torch(g, 10_000_000, 150, seed=7)
    del g
    torch.cuda.synchronize()
    with DeviceIndex.from_index(index) as dev:
        print("index %d k-mers, radix_min_units %d positions (= %d reads of 150 bp)"
              % (n_index, dev.get_param("radix_min_units"), dev.get_param("radix_min_units") // 150))
        print("%12s %14s %14s   (G k-mers/s)" % ("reads/call", "direct", "radix"))
        sizes = (20_000, 50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000, 3_200_000, 10_000_000)
        paths = (1, 2)
        if len(sys.argv) > 3:
            sizes, paths = (int(sys.argv[2]),), (int(sys.argv[3]),)
        for R in sizes:
            view = big[: R * 150]
            row = []
            for path in paths:
                dev.set_param("path", path)
                calls = max(3, min(300, 30_000_000 // R))
                for _ in range(2):
                    dev.map_reads_uniform(view, R, 150, 31)
                dev.synchronize()
                t0 = time.perf_counter()
                for _ in range(calls):
                    dev.map_reads_uniform(view, R, 150, 31)
                dev.synchronize()
                dt = time.perf_counter() - t0
                row.append(R * 120 * calls / dt / 1e9)
            print("%12d " % R + " ".join("%14.2f" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
