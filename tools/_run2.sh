set -u
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4b/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r4b/tests.log
[ $rc -ne 0 ] && exit 1
B="--no-cpu-baseline --no-h2d-leg"
for r in 10000000 20000000 30000000; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --reads $r 2>/dev/null | tail -1 > gpurun_out/r4b/cfg2_reads_$r.json
  python - $r <<'PY'
import json,sys
j=json.load(open("gpurun_out/r4b/cfg2_reads_%s.json"%sys.argv[1]))
print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["kernel_ms_per_step"], flush=True)
PY
done
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records 2>/dev/null | tail -1 > gpurun_out/r4b/cfg2_records.json
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4b/cfg2_records.json"))
print("records", j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
PY
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --config 1 2>/dev/null | tail -1 > gpurun_out/r4b/cfg1_records.json
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4b/cfg1_records.json"))
print("records cfg1", j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
PY
