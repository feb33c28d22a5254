set -u
mkdir -p gpurun_out/r4i
echo "== default"; timeout -k 10 300 python tools/records_stress.py 25 > gpurun_out/r4i/stress_default.txt 2>&1; tail -4 gpurun_out/r4i/stress_default.txt; grep -c FAILED gpurun_out/r4i/stress_default.txt
echo "== no overlap"; KMM_RECORDS_NO_OVERLAP=1 timeout -k 10 300 python tools/records_stress.py 25 > gpurun_out/r4i/stress_no_overlap.txt 2>&1; tail -2 gpurun_out/r4i/stress_no_overlap.txt; grep -c FAILED gpurun_out/r4i/stress_no_overlap.txt
echo "== full division in pass 1"; KMM_LIB_PATH=build_ab/libkmm_noinc.so timeout -k 10 300 python tools/records_stress.py 25 > gpurun_out/r4i/stress_noinc.txt 2>&1; tail -2 gpurun_out/r4i/stress_noinc.txt; grep -c FAILED gpurun_out/r4i/stress_noinc.txt
grep FAILED gpurun_out/r4i/*.txt | head -5
