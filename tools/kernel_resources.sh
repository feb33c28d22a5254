#!/bin/bash
# Register / scratch / LDS use of every kernel of libkmm (compile-only, no GPU):  tools/kernel_resources.sh [-DFLAG ...] [| grep k_rx]
# A kernel of the radix passes that spills inside its block / item loop loses ~30 %: check after every change.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude "$@" -Rpass-analysis=kernel-resource-usage \
    -o /tmp/libkmm_resources.so kmer_mapper_amd/csrc/kmm.hip 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r": remark: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-70s %5s %5s %7s %5s %6s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "vspill", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    print("%-70s %5s %5s %7s %5s %6s %7s" % (n[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
          r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
'
