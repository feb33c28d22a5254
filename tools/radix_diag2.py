import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.engine import DeviceIndex
from oracle import oracle
index, genome = syn.make_index(20000, seed=301)
mx = index.max_node_id()
bases, offs = syn.make_ragged_reads(genome, 30000, 0, 260, seed=302)
for shift in (12, 7):
    expect, n = oracle.map_reads(index, mx, bases, offs, 31, also_revcomp=True, n_threads=4)
    with DeviceIndex.from_index(index, mx) as dev:
        dev.set_param("part_shift", shift); dev.set_param("path", 2)
        print("F1", dev.get_param("n_coarse_partitions"), "PF", dev.get_param("n_partitions"))
        for r in range(4):
            dev.reset(); dev.get_stats(reset=True); dev.get_param("dbg0z")
            dev.map_reads(bases, offs, 31, also_revcomp=True)
            got = dev.get_node_counts()
            d = [dev.get_param("dbg%d" % i) for i in range(5)]
            print("shift", shift, "rep", r, "wrong", int((got != expect).sum()), "p2-lookups", dev.get_param("radix_p2_kmers") - dev.get_stats()[0],
                  "dups", d[0], "last dup (sub,idx)=(%d,%d) (cc,cj)=(%d,%d) item %d" % (d[1] >> 32, d[1] & 0xFFFFFFFF, d[2] >> 32, d[2] & 0xFFFFFFFF, d[3]), "processed", d[4], flush=True)
