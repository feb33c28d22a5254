#!/usr/bin/env python3
"""Randomised stress of the radix path against the oracle: index sizes, table sizes (also non-prime, also denser than
one entry per bucket), slice widths 0..13, ragged / uniform reads, reverse complements, k, the k-mer operator entry
point, several calls per handle.  Every case must be bit-exact and conserve its k-mers through the passes.

    python tools/radix_stress.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmer_mapper_amd import synthetic as syn                   # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex                 # noqa: E402
from oracle import oracle                                      # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    done = skipped = 0
    for case in range(cases):
        n_index = int(rng.choice([300, 2000, 20000, 150000]))
        load = float(rng.choice([0.2, 0.5, 1.0, 2.5]))
        modulo = max(3, int(n_index / load) + int(rng.integers(0, 7)))
        k = int(rng.choice([15, 21, 27, 31]))
        index, genome = syn.make_index(n_index, k=k, seed=int(rng.integers(1, 1 << 30)), modulo=modulo,
                                       skewed=bool(rng.integers(0, 2)))
        mx = index.max_node_id()
        n_reads = int(rng.choice([200, 5000, 40000]))
        uniform = bool(rng.integers(0, 2))
        if uniform:
            L = int(rng.choice([k, 64, 150, 251]))
            bases, offs = syn.make_reads(genome, n_reads, L, seed=int(rng.integers(1, 1 << 30)))
        else:
            bases, offs = syn.make_ragged_reads(genome, n_reads, 0, int(rng.choice([40, 300])), seed=int(rng.integers(1, 1 << 30)))
        rc = bool(rng.integers(0, 2))
        mf = int(rng.choice([1, 3, 1000]))
        expect, n = oracle.map_reads(index, mx, bases, offs, k, max_index_lookup_frequency=mf, also_revcomp=rc, n_threads=8)
        with DeviceIndex.from_index(index, mx) as dev:
            if dev.get_param("radix_available") != 1:
                skipped += 1
                continue
            shift = int(rng.integers(0, 14))
            try:
                dev.set_param("part_shift", shift)
            except Exception:
                shift = dev.get_param("part_shift")
            dev.set_param("path", 2)
            dev.get_stats(reset=True)
            if uniform and rng.integers(0, 2):
                dev.map_reads_uniform(bases, n_reads, L, k, mf, also_revcomp=rc)
            else:
                dev.map_reads(bases, offs, k, mf, also_revcomp=rc)
            got = dev.get_node_counts()
            lookups = (2 if rc else 1) * n
            ok = np.array_equal(got, expect) and dev.get_param("radix_p2_kmers") == lookups \
                and dev.get_param("radix_p3_kmers") + dev.get_param("radix_p2_dropped") == lookups
            # the operator entry point on top (counts accumulate)
            km = oracle.extract(bases, offs, k)
            dev.map_kmers(km, mf, also_revcomp=rc, k=k)
            ok = ok and np.array_equal(dev.get_node_counts(), expect + expect)
            print("case %2d: N=%6d M=%7d k=%2d shift=%2d reads=%5d %s rc=%d mf=%4d skew=%d -> %s" % (
                case, n_index, modulo, k, shift, n_reads, "uniform" if uniform else "ragged ", rc, mf,
                int(mx < n_index - 1), "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                sys.exit(1)
            done += 1
    print("radix stress: %d cases bit-exact, %d without a radix path, %.1f s" % (done, skipped, time.time() - t0))


if __name__ == "__main__":
    main()
