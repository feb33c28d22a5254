#!/usr/bin/env python3
"""Does the radix path stay bit-exact while OTHER kernels run beside it on another stream?  Rounds of two back-to-back
uniform map calls (reads resident in HBM) with a torch side stream kept busy by element-wise kernels on its own tensors;
every round ends with a synchronising call (conservation self-check) and a comparison with round 0.
    python tools/concurrency_stress.py [rounds=20] [mode=torch|none]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "torch"
    R, L, k = 10_000_000, 150, 31
    index, genome = syn.make_index(100_000_000, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    g = torch.from_numpy(syn.ACGT[genome]).cuda()
    batches = [syn.make_reads_torch(g, R, L, seed=1000 + b) for b in range(2)]
    del g
    side = torch.cuda.Stream()
    junk = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    first, fails = None, 0
    with DeviceIndex.from_index(index, mx) as dev:
        for r in range(rounds):
            dev.reset()
            try:
                for b in range(2):
                    if mode == "torch":
                        with torch.cuda.stream(side):
                            for _ in range(12):
                                junk.add_(1)            # ~0.1 ms kernels on the side stream while the passes run
                    dev.map_reads_uniform(batches[b], R, L, k)
                got = dev.get_node_counts()
            except Exception as e:      # noqa: BLE001
                fails += 1
                print("round %d FAILED: %s" % (r, str(e)[:260]), flush=True)
                continue
            if first is None:
                first = got
            same = np.array_equal(got, first)
            fails += 0 if same else 1
            print("round %d: %s" % (r, "same as round 0" if same else "DIFFERENT"), flush=True)
            side.synchronize()
    print("failures: %d of %d (%s)" % (fails, rounds, mode))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
