set -u
mkdir -p gpurun_out/r4f
B="--no-cpu-baseline --no-h2d-leg"
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --config 2 > gpurun_out/r4f/cfg2_records_$i.json 2> gpurun_out/r4f/cfg2_records_$i.err; echo "run $i rc=$?"; tail -3 gpurun_out/r4f/cfg2_records_$i.err
python - $i <<'PY'
import json,sys
try:
    j=json.loads(open("gpurun_out/r4f/cfg2_records_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
    print("records cfg2", j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
except Exception as e:
    print("no json", e)
PY
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KMM_RECORDS_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f/prof_records -- python3 bench.py --steps 4 --warmup 1 $B --records > gpurun_out/r4f/prof_records.json 2> gpurun_out/r4f/prof_records.err
find gpurun_out/r4f/prof_records -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4f/records_serial_kernel_stats.csv
rm -rf gpurun_out/r4f/prof_records
grep -E "k_rec|k_rx_p" gpurun_out/r4f/records_serial_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200 | sed 's/(anonymous namespace):://g' | cut -c1-60,120-200
