"""Phase breakdown of the radix passes from a -DRX_PHASE_TIMERS build (tools/ab_build.sh pt="-DRX_PHASE_TIMERS"
ptp1="-DRX_PHASE_TIMERS -DRX_PT_P1"):   KMM_LIB_PATH=build_ab/libkmm_pt.so python tools/phase_timers.py
Prints, per pass, the share of thread 0's shader-clock cycles spent in each phase, summed over the workgroups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kmer_mapper_amd import synthetic as syn
from kmer_mapper_amd.engine import DeviceIndex

n_index = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
R, L, k = 10_000_000, 150, 31
index, genome = syn.make_index(n_index, k=k, seed=1, gpu_builder=True)
mx = index.max_node_id()
g = torch.from_numpy(syn.ACGT[genome]).cuda()
reads = syn.make_reads_torch(g, R, L, seed=1001)
del g
with DeviceIndex.from_index(index, mx) as dev:
    dev.set_param("path", 2)
    dev.map_reads_uniform(reads, R, L, k)
    dev.synchronize()
    dev.get_stats(reset=True)
    dev.set_timing(True)
    dev.map_reads_uniform(reads, R, L, k)
    dev.synchronize()
    t = dev.get_timing()
    print({n: round(v[0] / max(v[1], 1), 3) for n, v in t.items() if v[1]})
    s = [dev.get_param("stats_slot_%d" % i) for i in range(16)]
    p1 = "ptp1" in os.environ.get("KMM_LIB_PATH", "")
    names3 = ["wait item", "slice load", "scan+list", "stream+probe", "flush", "-"]
    names2 = ["wait item", "descriptors+scan", "keys+ranks", "scan+place", "copy-out", "list+gather"]
    if dev.get_param("radix_filter"):      # k_rx_p2f's phases
        names2 = ["unit prologue / loop top", "table + search + requests", "keys+ranks", "scan+place", "copy-out",
                  "wait for requests + filter + compaction"]
    names1 = ["front end", "-", "keys+ranks (division)", "scan+place", "copy-out", "-"]
    for title, base, names in (("pass 3", 4, names3), ("pass 1" if p1 else "pass 2", 10, names1 if p1 else names2)):
        tot = sum(s[base:base + 6]) or 1
        print(title, "total cycles/workgroup-sum %.3e" % tot)
        for i in range(6):
            if s[base + i]:
                print("   %-24s %5.1f %%" % (names[i], 100.0 * s[base + i] / tot))
