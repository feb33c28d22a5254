#!/usr/bin/env python3
"""Mean counter value per kernel from rocprofv3 --pmc CSV output (counter_collection.csv files under a dir).
    python tools/pmc_summary.py DIR [substring-of-kernel-name ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2:] or ["k_rx_", "k_map_"]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if not any(w in name for w in want):
                continue
            import re
            m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
            short = (m.group(1) + (m.group(2) or "")) if m else name
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    for k in out:
        out[k]["_dispatches"] = len(next(iter(acc[k].values())))
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
