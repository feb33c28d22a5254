#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another: they do not fit one
pass on gfx950) into profiles/hbm_traffic.json, the per-launch HBM byte count bench.py reports as
roofline.traffic.

    python tools/pmc_traffic.py --fetch DIR/x_counter_collection.csv --write DIR/y_counter_collection.csv \
        --kernel k_map_reads --reads 10000000 --index-kmers 10000000 --stream-bytes 1.5e9 --out profiles/hbm_traffic.json

Units/corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B
(FETCH_SIZE = TCC_EA0_RDREQ x 64 B / 1024).  On gfx950 a wide coalesced streaming read is tallied at
half its bytes, so the streamed read bytes of the kernel (--stream-bytes: the read bases, 16 B/lane
loads) are added once more; the random 16-byte gathers are single 64-byte fabric requests and are taken
as counted.  WRITE_SIZE is taken as counted (atomics: one request per lane).
"""
import argparse
import csv
import json


def mean_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if not vals:
        raise SystemExit("no %s rows for kernel %s in %s" % (counter, kernel, path))
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--kernel", default="k_map_reads")
    ap.add_argument("--reads", type=int, required=True)
    ap.add_argument("--index-kmers", type=int, required=True)
    ap.add_argument("--stream-bytes", type=float, default=0.0)
    ap.add_argument("--tcc", default=None, help="optional pass with TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_sum")
    ap.add_argument("--kmers-per-launch", type=float, default=None)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch, nf = mean_counter(a.fetch, a.kernel, "FETCH_SIZE")
    write, nw = mean_counter(a.write, a.kernel, "WRITE_SIZE")
    total = fetch * 1024 + a.stream_bytes * 0.5 + write * 1024
    out = {
        "kernel": a.kernel, "reads": a.reads, "index_kmers": a.index_kmers,
        "FETCH_SIZE_per_launch": fetch, "WRITE_SIZE_per_launch": write, "launches_seen": [nf, nw],
        "stream_read_correction_bytes": a.stream_bytes * 0.5,
        "hbm_bytes_per_launch": total,
        "note": "FETCH_SIZE*1024 + half of the coalesced read stream again (gfx950 tallies 128-B stream "
                "requests at 64 B) + WRITE_SIZE*1024; separate --pmc passes",
    }
    if a.tcc:
        rd, _ = mean_counter(a.tcc, a.kernel, "TCC_EA0_RDREQ_sum")
        hit, _ = mean_counter(a.tcc, a.kernel, "TCC_HIT_sum")
        wr, _ = mean_counter(a.tcc, a.kernel, "TCC_EA0_WRREQ_sum")
        out["TCC_EA0_RDREQ_per_launch"] = rd
        out["TCC_HIT_per_launch"] = hit
        out["TCC_EA0_WRREQ_per_launch"] = wr
        if a.kmers_per_launch:
            out["l2_miss_read_requests_per_kmer"] = rd / a.kmers_per_launch
            out["l2_hits_per_kmer"] = hit / a.kmers_per_launch
            out["atomic_write_requests_per_kmer"] = wr / a.kmers_per_launch
    # the output file holds one entry per (kernel, reads, index_kmers); replace the matching one
    import os
    doc = {"note": "one entry per committed PMC pass, keyed by (kernel, reads per batch, index_kmers); bench.py "
                   "reports the matching entry as roofline.traffic", "entries": []}
    if os.path.exists(a.out):
        try:
            old = json.load(open(a.out))
            doc["entries"] = old.get("entries", [old])
        except Exception:
            pass
    key = (a.kernel, a.reads, a.index_kmers)
    doc["entries"] = [e for e in doc["entries"] if (e.get("kernel"), e.get("reads"), e.get("index_kmers")) != key]
    doc["entries"].append(out)
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
