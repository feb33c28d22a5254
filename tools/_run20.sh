set -u
mkdir -p gpurun_out/r4k
timeout -k 10 600 python tools/records_overlap_bisect.py 10 count2 > gpurun_out/r4k/overlap_bisect_sums.txt 2>&1; tail -12 gpurun_out/r4k/overlap_bisect_sums.txt
