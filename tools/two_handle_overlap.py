"""Do the radix passes of two batches overlap usefully on one GPU?  Two handles of the same index (each with its own
stream and buffers) map alternate batches; compared with one handle mapping all of them.  A ceiling for pipelining the
passes of consecutive batches inside one handle (pass 1 of batch i + 1 beside pass 3 of batch i).
    python tools/two_handle_overlap.py [n_index] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.engine import DeviceIndex

n_index = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R, L, k = 10_000_000, 150, 31
index, genome = syn.make_index(n_index, k=k, seed=1, gpu_builder=True)
mx = index.max_node_id()
g = torch.from_numpy(syn.ACGT[genome]).cuda()
batches = [syn.make_reads_torch(g, R, L, seed=1001 + i) for i in range(2)]
del g


def run(handles, n):
    for h in handles:
        h.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        handles[i % len(handles)].map_reads_uniform(batches[i & 1], R, L, k)
    for h in handles:
        h.synchronize()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


with DeviceIndex.from_index(index, mx) as a, DeviceIndex.from_index(index, mx) as b:
    for h in (a, b):
        h.set_param("path", 2)
    run([a, b], 4)
    for name, hs in (("one handle", [a]), ("two handles", [a, b]), ("one handle", [a]), ("two handles", [a, b])):
        t = run(hs, steps)
        print("%-12s %7.3f ms per batch  %6.1f G k-mers/s" % (name, t / steps * 1e3, steps * R * (L - k + 1) / t / 1e9), flush=True)
    ca, cb = a.get_node_counts(), b.get_node_counts()
    print("counts: handle a %d, handle b %d hits" % (int(ca.sum()), int(cb.sum())))
