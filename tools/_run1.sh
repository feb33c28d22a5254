set -u
mkdir -p gpurun_out/r4a
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r4a/tests.log
timeout -k 10 300 python tools/ab_run.py base inc > gpurun_out/r4a/ab_cfg2.txt 2>&1; cat gpurun_out/r4a/ab_cfg2.txt
timeout -k 10 300 python tools/ab_run.py base inc -- --config 1 > gpurun_out/r4a/ab_cfg1.txt 2>&1; cat gpurun_out/r4a/ab_cfg1.txt
timeout -k 10 300 python tools/ab_run.py base inc -- --general-path > gpurun_out/r4a/ab_cfg2_general.txt 2>&1; cat gpurun_out/r4a/ab_cfg2_general.txt
timeout -k 10 500 python bench.py --index-kmers 1000000000 --reads 28000000 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d-leg > gpurun_out/r4a/bench_1B_28M.json 2> gpurun_out/r4a/bench_1B_28M.err; echo "1B rc=$?"; tail -c 1500 gpurun_out/r4a/bench_1B_28M.json
