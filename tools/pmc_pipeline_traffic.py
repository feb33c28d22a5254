#!/usr/bin/env python3
"""HBM traffic per bench step from separate rocprofv3 --pmc passes -> one entry of profiles/hbm_traffic.json.

    python tools/pmc_pipeline_traffic.py --dir gpurun_out/pmc_cfg2 --kernels k_rx_p1,k_rx_colsum,... \
        --steps-kernel k_rx_p1 --key "k_rx_p1+k_rx_scan+k_rx_p2+k_rx_p3" --reads 10000000 --index-kmers 100000000 \
        --stream-read-bytes 1.5e9 --out profiles/hbm_traffic.json

--dir holds one sub-directory per pass (fetch/, write/, rdsize/, wrsize/), each with rocprofv3's
*counter_collection.csv.  Counters are summed over every dispatch of the named kernels and divided by the number
of steps (= dispatches of --steps-kernel): the pipeline of one bench step is one launch of each kernel.

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in units of 1024 bytes and
derive from the L2's fabric-side request counters; on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B, i.e. a 128-byte
request of a wide coalesced stream is tallied at half its size.  Instead of guessing which reads were 128-byte
requests, the read side is priced from the size-specific request counters of a further pass,
32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B (they add up to RDREQ); the guide's rule "FETCH_SIZE x 2
for coalesced streams" is the special case RDREQ_128B = RDREQ.  WRITE_SIZE is taken as counted (64-byte and
32-byte write requests are both tallied at their size: WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32 is reported next
to it as a cross-check).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def load(pass_dir):
    """{kernel short name: {counter: [values per dispatch]}}"""
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--kernels", required=True, help="comma-separated kernel names (exact, without template args)")
    ap.add_argument("--steps-kernel", required=True)
    ap.add_argument("--key", required=True, help="value of roofline.kernel in bench.py's line for this path")
    ap.add_argument("--reads", type=int, required=True)
    ap.add_argument("--index-kmers", type=int, required=True)
    ap.add_argument("--kmers-per-step", type=float, default=None)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    kernels = a.kernels.split(",")
    passes = {name: load(os.path.join(a.dir, name)) for name in ("fetch", "write", "rdsize", "wrsize")}

    def per_step(pass_name, counter):
        d = passes[pass_name]
        steps = len(d[a.steps_kernel][counter])
        if not steps:
            raise SystemExit("no %s values for %s in pass %s" % (counter, a.steps_kernel, pass_name))
        return sum(sum(d[k][counter]) for k in kernels if k in d) / steps, steps

    fetch, n1 = per_step("fetch", "FETCH_SIZE")
    write, n2 = per_step("write", "WRITE_SIZE")
    rd = {c: per_step("rdsize", c)[0] for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum",
                                               "TCC_EA0_RDREQ_128B_sum")}
    wr = {c: per_step("wrsize", c)[0] for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_HIT_sum", "TCC_MISS_sum")}
    read_bytes = 32 * rd["TCC_EA0_RDREQ_32B_sum"] + 64 * rd["TCC_EA0_RDREQ_64B_sum"] + 128 * rd["TCC_EA0_RDREQ_128B_sum"]
    write_bytes = write * 1024
    per_kernel = {}
    for k in kernels:
        if k in passes["rdsize"]:
            d, w = passes["rdsize"][k], passes["write"].get(k, {})
            n = max(len(passes["rdsize"][a.steps_kernel]["TCC_EA0_RDREQ_sum"]), 1)
            rb = (32 * sum(d["TCC_EA0_RDREQ_32B_sum"]) + 64 * sum(d["TCC_EA0_RDREQ_64B_sum"])
                  + 128 * sum(d["TCC_EA0_RDREQ_128B_sum"])) / n
            wb = sum(w.get("WRITE_SIZE", [])) * 1024 / max(n2, 1)
            per_kernel[k] = {"read_bytes": rb, "write_bytes": wb}
    out = {
        "kernel": a.key, "reads": a.reads, "index_kmers": a.index_kmers,
        "FETCH_SIZE_per_step": fetch, "WRITE_SIZE_per_step": write, "steps_seen": [n1, n2],
        "TCC_EA0_RDREQ_per_step": rd, "TCC_EA0_WRREQ_per_step": wr,
        "read_bytes_per_step": read_bytes, "write_bytes_per_step": write_bytes,
        "fetch_size_x1024": fetch * 1024,
        "write_bytes_from_request_sizes": 64 * wr["TCC_EA0_WRREQ_64B_sum"] + 32 * (wr["TCC_EA0_WRREQ_sum"] - wr["TCC_EA0_WRREQ_64B_sum"]),
        "hbm_bytes_per_launch": read_bytes + write_bytes,
        "l2_hit_rate": wr["TCC_HIT_sum"] / max(wr["TCC_HIT_sum"] + wr["TCC_MISS_sum"], 1.0),
        "per_kernel": per_kernel,
        "note": "separate --pmc passes (FETCH_SIZE; WRITE_SIZE; RDREQ by size; WRREQ/HIT/MISS); read side = 32/64/128 B x "
                "the size-specific fabric read requests (FETCH_SIZE tallies every request at 64 B on gfx950), write side = "
                "WRITE_SIZE x 1024",
    }
    if a.kmers_per_step:
        out["hbm_bytes_per_kmer"] = out["hbm_bytes_per_launch"] / a.kmers_per_step
    doc = {"note": "one entry per committed PMC pass, keyed by (kernel, reads per batch, index_kmers); bench.py "
                   "reports the matching entry as roofline.traffic", "entries": []}
    if os.path.exists(a.out):
        try:
            old = json.load(open(a.out))
            doc["entries"] = old.get("entries", [old])
        except Exception:
            pass
    key = (a.key, a.reads, a.index_kmers)
    doc["entries"] = [e for e in doc["entries"] if (e.get("kernel"), e.get("reads"), e.get("index_kmers")) != key]
    doc["entries"].append(out)
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
    print(json.dumps(per_kernel, indent=1))


if __name__ == "__main__":
    main()
