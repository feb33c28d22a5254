set -u
mkdir -p gpurun_out/r4k
timeout -k 10 900 python tools/records_overlap_bisect.py 12 > gpurun_out/r4k/overlap_bisect.txt 2>&1; cat gpurun_out/r4k/overlap_bisect.txt | tail -24
