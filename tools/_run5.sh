set -u
mkdir -p gpurun_out/r4e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r4e/tests.log
[ $rc -ne 0 ] && exit 1
B="--no-cpu-baseline --no-h2d-leg"
for cfg in 2 1; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --config $cfg 2>/dev/null | tail -1 > gpurun_out/r4e/cfg${cfg}_records.json
python - $cfg <<'PY'
import json,sys
j=json.load(open("gpurun_out/r4e/cfg%s_records.json"%sys.argv[1]))
print("records cfg", sys.argv[1], j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
PY
done
timeout -k 10 600 python tools/gz_inflate_rate.py 4000000 /tmp/kmm_gz > gpurun_out/r4e/gz_inflate_rate.txt 2>&1; grep -E "GB/s|FASTQ" gpurun_out/r4e/gz_inflate_rate.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e/prof_records -- python3 bench.py --steps 4 --warmup 1 $B --records > gpurun_out/r4e/prof_records.json 2> gpurun_out/r4e/prof_records.err
find gpurun_out/r4e/prof_records -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4e/records_kernel_stats.csv
rm -rf gpurun_out/r4e/prof_records
grep -E "k_rec|k_rx_p" gpurun_out/r4e/records_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
