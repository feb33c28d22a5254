#!/usr/bin/env python3
"""Full-size determinism soak: a 10 M-read batch against the 100 M-k-mer index, 40 times on the radix path (identical
count vectors every time), then once on the direct path (the same vector)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.engine import DeviceIndex
index, genome = syn.make_index(100_000_000, k=31, seed=1, gpu_builder=True)
mx = index.max_node_id()
g = torch.from_numpy(syn.ACGT[genome]).cuda()
R, L, k = 10_000_000, 150, 31
reads = syn.make_reads_torch(g, R, L, seed=1001)
del g
with DeviceIndex.from_index(index, mx) as dev:
    dev.set_param("path", 2)
    ref = None
    t0 = time.perf_counter()
    for rep in range(40):
        dev.reset()
        dev.map_reads_uniform(reads, R, L, k)
        c = dev.get_node_counts()
        if ref is None:
            ref = c.copy()
        else:
            assert np.array_equal(c, ref), "repetition %d differs" % rep
    print("40 repetitions of a 10 M-read batch on the radix path: identical counts (%d hits), %.1f s" % (int(ref.astype(np.uint64).sum()), time.perf_counter() - t0))
    # direct path gives the same vector
    dev.reset(); dev.set_param("path", 1); dev.map_reads_uniform(reads, R, L, k)
    assert np.array_equal(dev.get_node_counts(), ref)
    print("direct path: same counts")
