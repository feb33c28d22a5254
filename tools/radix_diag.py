#!/usr/bin/env python3
"""Diagnostic for the radix path: conservation of k-mers through the passes and parity against the oracle on a
small ragged-read case, repeated.  python tools/radix_diag.py [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_mapper_amd import synthetic as syn            # noqa: E402
from kmer_mapper_amd.engine import DeviceIndex          # noqa: E402
from oracle import oracle                               # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
index, genome = syn.make_index(20000, seed=301)
mx = index.max_node_id()
bases, offs = syn.make_ragged_reads(genome, 30000, 0, 260, seed=302)
for shift in (7, 12, 4):
    for rc in (True, False):
        expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=rc, n_threads=4)
        with DeviceIndex.from_index(index, mx) as dev:
            dev.set_param("part_shift", shift)
            dev.set_param("path", 2)
            for r in range(reps):
                dev.reset()
                dev.get_stats(reset=True)
                dev.map_reads(bases, offs, 31, also_revcomp=rc)
                got = dev.get_node_counts()
                lk, hits = dev.get_stats()
                p2, p3 = dev.get_param("radix_p2_kmers"), dev.get_param("radix_p3_kmers")
                bad = int((got != expect).sum())
                print("shift %2d rc %d rep %d: lookups %d (expect %d) p2 %d p3 %d hits %d (expect %d) wrong nodes %d diff %d"
                      % (shift, rc, r, lk, (2 if rc else 1) * n, p2, p3, hits, int(expect.sum()), bad,
                         int(got.astype(np.int64).sum() - expect.astype(np.int64).sum())), flush=True)
