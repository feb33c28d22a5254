set -u
mkdir -p gpurun_out/r4d
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_radix.py -x -q > gpurun_out/r4d/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r4d/tests.log
[ $rc -ne 0 ] && exit 1
# plain gzip on the box's cores
timeout -k 10 600 python tools/gz_inflate_rate.py 2000000 /tmp/kmm_gz --no-cli > gpurun_out/r4d/gz_inflate_rate.txt 2>&1; tail -8 gpurun_out/r4d/gz_inflate_rate.txt
# N = 2 rehearsal over gloo (two ranks share the GPU): weak value + strong + staged legs in one invocation
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --dist-backend gloo > gpurun_out/r4d/bench_2rank_gloo.json 2> gpurun_out/r4d/bench_2rank_gloo.err; echo "2rank rc=$?"; tail -c 3000 gpurun_out/r4d/bench_2rank_gloo.json
