#!/usr/bin/env python3
"""Stress of kmm_map_records on the radix path at configs[2] size: N rounds of two back-to-back 10 M-read FASTQ calls (the
second call's compaction kernels run on the copy stream beside the first call's passes) followed by a synchronising
call that runs the conservation self-check; counts are compared with the first round's.
    python tools/records_stress.py [rounds=20] [n_index=100000000] [reads=10000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from kmer_mapper_amd import _lib, synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_index = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    L, k = 150, 31
    index, genome = syn.make_index(n_index, k=k, seed=1, gpu_builder=True)
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
    first, fails = None, 0
    with DeviceIndex.from_index(index, mx) as dev:
        # KMM_RECORDS_COPY_STREAM=1: the variant that exposed the counter-clear race of round 3's sort — the second call
        # only runs its compaction kernels, on the copy stream, beside the first call's radix passes (it maps nothing:
        # "debug_records_copy_stream"); the counts must be those of the first call alone
        beside = bool(os.environ.get("KMM_RECORDS_COPY_STREAM"))
        if beside:
            print("second call = compaction kernels only, on the copy stream, beside the first call's radix passes")
        for r in range(rounds):
            dev.reset()
            t = time.perf_counter()
            try:
                for b in range(2):
                    if beside:
                        dev.set_param("debug_records_copy_stream", b)
                    used, n_rec = dev.map_records(fq[b], fmt=_lib.FORMAT_FASTQ, k=k)
                    assert n_rec == R and used == fq[b].numel()
                if beside:
                    dev.set_param("debug_records_copy_stream", 0)
                got = dev.get_node_counts()
            except Exception as e:      # noqa: BLE001
                fails += 1
                print("round %d FAILED: %s" % (r, str(e)[:300]), flush=True)
                continue
            dt = time.perf_counter() - t
            if first is None:
                first = got
            same = np.array_equal(got, first)
            fails += 0 if same else 1
            print("round %d: %.1f ms, %d hits, %s" % (r, dt * 1e3, int(got.astype(np.uint64).sum()),
                                                     "same as round 0" if same else "DIFFERENT"), flush=True)
    print("failures: %d of %d" % (fails, rounds))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
