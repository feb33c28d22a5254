#!/usr/bin/env python3
"""Which of the compaction kernels of kmm_map_records disturbs the radix passes when it runs BESIDE them?  (Round-4
fault: profiles/r04/records_overlap_fault.txt.)  Rounds of: one normal records call (its passes are enqueued on the
handle's stream), then at once a second call whose compaction runs on the COPY stream with a subset of its kernels
(debug_records_copy_stream / debug_records_skip; that call maps nothing), then a synchronising call = the conservation
self-check of the first call's passes.     python tools/records_overlap_bisect.py [rounds=12]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from kmer_mapper_amd import _lib, synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    no_filter = len(sys.argv) > 2 and sys.argv[2] == "nofilter"     # pass 2 without its filter: the variant without scratch
    only = len(sys.argv) > 2 and sys.argv[2] == "count2"            # just the variant that fails, with the directory sums
    R, L, k = 10_000_000, 150, 31
    index, genome = syn.make_index(100_000_000, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    g = torch.from_numpy(syn.ACGT[genome]).cuda()
    fq = []
    for b in range(2):
        reads = syn.make_reads_torch(g, R, L, seed=1000 + b)
        rec = torch.empty((R, 4 + L + 3 + L + 1), dtype=torch.uint8, device="cuda")
        rec[:, 0:4] = torch.tensor(list(b"@rd\n"), dtype=torch.uint8, device="cuda")
        rec[:, 4:4 + L] = reads.view(R, L)
        rec[:, 4 + L:7 + L] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device="cuda")
        rec[:, 7 + L:7 + 2 * L] = ord("F")
        rec[:, -1] = 10
        fq.append(rec.reshape(-1))
        del reads
    del g
    torch.cuda.synchronize()
    names = {1: "count2", 2: "scans", 4: "scatter", 8: "uniform", 16: "memsets"}
    variants = [("all kernels beside the passes", 0), ("nothing but the small memsets", 31), ("only the large memsets", 15),
                ("only count2", 30), ("only count2 + scans", 28), ("all but scatter", 4), ("all but count2", 1),
                ("only scatter (stale tables)", 27), ("all but the large memsets", 16)]
    with DeviceIndex.from_index(index, mx) as dev:
        if no_filter:
            dev.set_param("radix_filter", 0)
            variants = [v for v in variants if v[1] in (30, 1)]
            print("radix_filter = 0: k_rx_p2f<false, ...> (no scratch)", flush=True)
        if only:
            variants = [v for v in variants if v[1] == 30]
        for label, skip in variants:
            fails = 0
            for r in range(rounds):
                dev.reset()
                dev.set_param("debug_records_copy_stream", 0)
                dev.set_param("debug_records_skip", 0)
                dev.map_records(fq[0], fmt=_lib.FORMAT_FASTQ, k=k)
                dev.set_param("debug_records_copy_stream", 1)
                dev.set_param("debug_records_skip", skip)
                dev.map_records(fq[1], fmt=_lib.FORMAT_FASTQ, k=k)
                try:
                    dev.get_node_counts()
                except Exception as e:      # noqa: BLE001
                    fails += 1
                    if fails <= 2:
                        print("   failure:", str(e)[40:170], flush=True)
                        print("   the failing call's directory: blocks account for %d k-mers, coarse partitions for %d, %d items"
                              % (dev.get_param("debug_rx_start1_sum"), dev.get_param("debug_rx_t1_sum"),
                                 dev.get_param("debug_rx_items")), flush=True)
                dev.reset()                 # (clears the sticky error)
                dev.get_stats(reset=True)   # (and the passes' counters)
            left = [n for b, n in names.items() if not skip & b]
            print("%-34s (runs: %s): %d of %d rounds fail the self-check" % (label, ", ".join(left) or "-", fails, rounds), flush=True)


if __name__ == "__main__":
    main()
