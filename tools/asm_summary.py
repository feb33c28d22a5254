"""Instruction mix and scratch traffic of one kernel from a hipcc --save-temps assembly listing (compile-only, no GPU).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude --save-temps -o /tmp/asm/libkmm.so kmer_mapper_amd/csrc/kmm.hip   (in /tmp/asm)
    python tools/asm_summary.py /tmp/asm/kmm-hip-amdgcn-amd-amdhsa-gfx950.s 'k_rx_p1ILi4ELb0E'
Prints the labels (basic blocks) that hold scratch loads / stores, and the op histogram of the whole kernel."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*%s\S*:" % re.escape(pat), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
cnt = collections.Counter()
label = "entry"
per_label = collections.OrderedDict()
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            label = m.group(1)
        continue
    op = t.split()[0]
    cnt[op] += 1
    d = per_label.setdefault(label, collections.Counter())
    d[op] += 1
print("%d instructions" % sum(cnt.values()))
for lab, d in per_label.items():
    sc = {k: v for k, v in d.items() if "scratch" in k}
    if sc:
        print("  %-14s %5d instr  scratch: %s" % (lab, sum(d.values()), sc))
groups = collections.Counter()
for k, v in cnt.items():
    g = ("valu_mul32" if re.match(r"v_(mul_lo_u32|mul_hi_u32|mad_u64_u32|mul_lo_i32|mul_hi_i32|mad_i64_i32)", k) else
         "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
         "vmem" if re.match(r"(global|buffer|flat|scratch)_", k) else "other")
    groups[g] += v
print(dict(groups))
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:45]:
    print("%6d %s" % (v, k))
