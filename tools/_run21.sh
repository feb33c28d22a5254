set -u
mkdir -p gpurun_out/r4k
timeout -k 10 600 python tools/records_overlap_bisect.py 16 count2 > gpurun_out/r4k/overlap_bisect_after_fix.txt 2>&1; tail -6 gpurun_out/r4k/overlap_bisect_after_fix.txt
timeout -k 10 600 python tools/records_overlap_bisect.py 8 > gpurun_out/r4k/overlap_bisect_after_fix_all.txt 2>&1; tail -12 gpurun_out/r4k/overlap_bisect_after_fix_all.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4k/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r4k/tests.log
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg --reads 10000000 2>/dev/null | tail -1 > gpurun_out/r4k/cfg2_10M_after_fix.json
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4k/cfg2_10M_after_fix.json"))
print("cfg2 10M", j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"])
PY
