set -u
mkdir -p gpurun_out/r4k
echo "== records stress, kernels serialised on the handle's stream"; echo skipped
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4k/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r4k/tests.log
[ $rc -ne 0 ] && exit 1
B="--no-cpu-baseline --no-h2d-leg"
for cfg in 2 1; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --reads 10000000 --config $cfg > gpurun_out/r4k/final_cfg${cfg}_records.json 2> gpurun_out/r4k/cfg${cfg}_records.err
python - $cfg <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r4k/final_cfg%s_records.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("records cfg", sys.argv[1], j["value"], j["ms_per_step"], j["config"]["kernel_ms_per_step"], flush=True)
PY
done
timeout -k 10 900 python tools/cli_e2e_large.py 10000000 100000000 /tmp/kmm_e2e_large > gpurun_out/r4k/cli_e2e_large_fastq.txt 2>&1; grep -E "E2E|counts vs|accumulated" gpurun_out/r4k/cli_e2e_large_fastq.txt | tail -5
timeout -k 10 600 python tools/gz_inflate_rate.py 4000000 /tmp/kmm_gz > gpurun_out/r4k/gz_inflate_rate.txt 2>&1; grep -E "GB/s" gpurun_out/r4k/gz_inflate_rate.txt | tail -12
