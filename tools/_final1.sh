set -u
O=gpurun_out/r4final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/final_default_bench.json 2> $O/final_default_bench.err; echo "default rc=$?"
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --reads 10000000 > $O/final_cfg2_10M_read_batches.json 2> /dev/null; echo "cfg2-10M rc=$?"
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --config 1 > $O/final_cfg1_bench.json 2> /dev/null; echo "cfg1 rc=$?"
B="--no-cpu-baseline --no-h2d-leg"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --reads 10000000 > $O/final_cfg2_records.json 2>/dev/null; echo "rec2 rc=$?"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --records --config 1 > $O/final_cfg1_records.json 2>/dev/null; echo "rec1 rc=$?"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --general-path --reads 10000000 > $O/final_cfg2_general.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --operator --config 1 > $O/final_cfg1_operator.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 6 --warmup 2 $B --modulo 452930477 --reads 10000000 > $O/final_m452.json 2>/dev/null
timeout -k 10 800 python bench.py --index-kmers 1000000000 --reads 28000000 --steps 12 --warmup 1 $B > $O/bench_index_1B_radix.json 2> $O/bench_index_1B_radix.err; echo "1B rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4final/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %9.1f G/s %8.3f ms/step frac %.4f h2d %s cpu %s parity %s" % (f.split("/")[-1], j["value"]/1e3, j["ms_per_step"], j["roofline"]["frac"], j.get("value_incl_h2d"), (j.get("cpu_baseline") or {}).get("value"), j.get("parity_vs_oracle_on_sample")), j["config"]["kernel_ms_per_step"], flush=True)
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 900 python tools/cli_e2e_large.py 10000000 100000000 /tmp/kmm_e2e_large > $O/cli_e2e_large_fastq.txt 2>&1; grep -E "E2E|counts vs|path_taken|accumulated" $O/cli_e2e_large_fastq.txt | tail -8
