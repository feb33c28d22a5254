"""Do the radix path's passes overlap usefully when two streams run them side by side?

Two handles on the same GPU (each with its own stream and scratch buffers) map half-batches from two host threads;
the aggregate rate is compared with one handle mapping whole batches.  `radix_grid_per_cu` = 1 makes the persistent
workgroups of passes 2 and 3 leave one workgroup slot per CU to the other stream's kernels.

    python tools/overlap_probe.py [--index-kmers N] [--reads R] [--steps K]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--index-kmers", type=int, default=100_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--splits", type=int, default=2, help="pieces per batch and handle in the two-handle runs")
    args = ap.parse_args()
    import torch
    from kmer_mapper_amd import synthetic as syn
    from kmer_mapper_amd.engine import DeviceIndex
    k, L, R = 31, 150, args.reads
    index, genome = syn.make_index(args.index_kmers, k=k, seed=1, gpu_builder=True)
    mx = index.max_node_id()
    g_ascii = torch.from_numpy(syn.ACGT[genome]).cuda()
    reads = syn.make_reads_torch(g_ascii, R, L, seed=1001)
    del g_ascii
    torch.cuda.synchronize()
    n_kmers = R * (L - k + 1)

    def run(handles, pieces, grid):
        for h in handles:
            h.set_param("path", 2)
            h.set_param("radix_grid_per_cu", grid)
            h.reset()
        n_h = len(handles)
        per = R // (n_h * pieces)

        def worker(hi, steps):
            h = handles[hi]
            for _ in range(steps):
                for p in range(pieces):
                    a = (hi * pieces + p) * per
                    h.map_reads_uniform(reads[a * L:(a + per) * L], per, L, k)
            h.synchronize()

        def go(steps):
            th = [threading.Thread(target=worker, args=(i, steps)) for i in range(n_h)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        go(2)
        dt = go(args.steps)
        done = per * pieces * n_h * (L - k + 1) * args.steps
        return done / dt / 1e9, dt / args.steps * 1e3

    a = DeviceIndex.from_index(index, mx)
    b = DeviceIndex.from_index(index, mx)
    ref = None
    for name, handles, pieces, grid in (("one handle, whole batch, 2 wg/CU", [a], 1, 2),
                                        ("one handle, whole batch, 1 wg/CU", [a], 1, 1),
                                        ("one handle, %d pieces, 2 wg/CU" % (2 * args.splits), [a], 2 * args.splits, 2),
                                        ("two handles, %d pieces each, 2 wg/CU" % args.splits, [a, b], args.splits, 2),
                                        ("two handles, %d pieces each, 1 wg/CU" % args.splits, [a, b], args.splits, 1)):
        rate, ms = run(handles, pieces, grid)
        print("%-44s %7.1f G k-mers/s  %6.2f ms/step" % (name, rate, ms), flush=True)
    # same node counts from both ways of running (sum of the two handles' vectors = one handle's)
    a.reset(); b.reset()
    a.set_param("radix_grid_per_cu", 2)
    a.map_reads_uniform(reads, R, L, k)
    whole = a.get_node_counts().copy()
    a.reset()
    h = R // 2
    a.set_param("radix_grid_per_cu", 1); b.set_param("radix_grid_per_cu", 1)
    t1 = threading.Thread(target=lambda: a.map_reads_uniform(reads[:h * L], h, L, k))
    t2 = threading.Thread(target=lambda: b.map_reads_uniform(reads[h * L:], R - h, L, k))
    t1.start(); t2.start(); t1.join(); t2.join()
    import numpy as np
    s = a.get_node_counts() + b.get_node_counts()
    print("counts agree:", bool(np.array_equal(s, whole)), "k-mers", n_kmers)


if __name__ == "__main__":
    main()
