import sys, time
sys.path.insert(0, '.')
import numpy as np
from kmer_mapper_amd.kmer_index import KmerIndex
from kmer_mapper_amd.engine import build_index
for n, M in [(40000, 5_000_011), (3_000_000, 6_000_011), (10_000_000, 20_000_003)]:
    rng = np.random.default_rng(1)
    kmers = rng.integers(0, 2**62, size=n, dtype=np.uint64)
    nodes = rng.integers(0, 2**31-1, size=n)
    t=time.time(); a = KmerIndex.from_flat_kmers(kmers, nodes, M); t1=time.time()-t
    t=time.time(); b = build_index(kmers, nodes, M); t2=time.time()-t
    t=time.time(); b = build_index(kmers, nodes, M); t3=time.time()-t
    print(n, M, "numpy %.2fs gpu first %.2fs gpu second %.2fs" % (t1, t2, t3), flush=True)
