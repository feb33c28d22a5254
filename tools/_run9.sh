set -u
mkdir -p gpurun_out/r4h
B="--no-cpu-baseline --no-h2d-leg"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KMM_RECORDS_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4h/prof -- python3 bench.py --steps 3 --warmup 1 $B --records --reads 10000000 > gpurun_out/r4h/prof.json 2> gpurun_out/r4h/prof.err
f=$(find gpurun_out/r4h/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
out=[]
for r in rows:
    n=r["Kernel_Name"]
    if "k_rec" in n or "k_rx_p" in n or "fillBuffer" in n or "copyBuffer" in n:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        if t0 is None: t0=s
        short=n.replace("(anonymous namespace)::","").split("(")[0][:28]
        out.append("%10.3f ms  +%9.1f us  %s grid=%s" % ((s-t0)/1e6,(e-s)/1e3,short,r.get("Grid_Size_X","?")))
open("gpurun_out/r4h/records_trace.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[-70:]))
PY
rm -rf gpurun_out/r4h/prof
