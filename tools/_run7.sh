set -u
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_gpu_radix.py tests/test_gpu_parity.py -x -q -k "record or fastq or cli or golden" > gpurun_out/r4g/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r4g/tests.log
[ $rc -ne 0 ] && exit 1
B="--no-cpu-baseline --no-h2d-leg"
for cfg in 2 1; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --reads 10000000 --config $cfg > gpurun_out/r4g/cfg${cfg}_records.json 2> gpurun_out/r4g/cfg${cfg}_records.err
python - $cfg <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r4g/cfg%s_records.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("records cfg", sys.argv[1], j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KMM_RECORDS_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4g/prof_records -- python3 bench.py --steps 4 --warmup 1 $B --records > gpurun_out/r4g/prof_records.json 2> gpurun_out/r4g/prof_records.err
find gpurun_out/r4g/prof_records -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4g/records_serial_kernel_stats.csv
rm -rf gpurun_out/r4g/prof_records
grep -E "k_rec_sc|k_rec_co|k_rec_un" gpurun_out/r4g/records_serial_kernel_stats.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,40), $2}' 
