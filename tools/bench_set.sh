set -u
run() { name=$1; shift; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d-leg "$@" 2>/dev/null | tail -1 > gpurun_out/r02b_$name.json; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
j=json.loads(open("gpurun_out/r02b_%s.json"%n).read())
print("%-22s %8.1f G/s %7.3f ms %s" % (n, j["value"]/1e3, j["ms_per_step"], j["config"]["kernel_ms_per_step"]), flush=True)
PY
}
run cfg2
run cfg1 --config 1
run cfg1_skewed --config 1 --skewed
run cfg2_skewed --skewed
run cfg1_operator --config 1 --operator
run cfg1_general --config 1 --general-path
run cfg2_general --general-path
run cfg1_records_radix --config 1 --records --reads 3400000 --path 2
run cfg1_records_direct --config 1 --records --reads 3400000
