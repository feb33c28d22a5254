set -u
mkdir -p gpurun_out/r4j
echo "== uniform calls beside torch kernels on a side stream"; timeout -k 10 300 python tools/concurrency_stress.py 25 torch > gpurun_out/r4j/conc_torch.txt 2>&1; tail -2 gpurun_out/r4j/conc_torch.txt; grep FAILED gpurun_out/r4j/conc_torch.txt | head -3
echo "== same without the side stream"; timeout -k 10 300 python tools/concurrency_stress.py 25 none > gpurun_out/r4j/conc_none.txt 2>&1; tail -1 gpurun_out/r4j/conc_none.txt
