set -u
mkdir -p gpurun_out/r4k
timeout -k 10 600 python tools/records_overlap_bisect.py 12 nofilter > gpurun_out/r4k/overlap_bisect_nofilter.txt 2>&1; tail -8 gpurun_out/r4k/overlap_bisect_nofilter.txt
