set -u
bash tools/collect_profiles.sh cfg2 > gpurun_out/collect_cfg2.log 2>&1; echo "cfg2 profiles rc=$?"; tail -3 gpurun_out/collect_cfg2.log
bash tools/collect_sq_counters.sh cfg2 > gpurun_out/collect_sq_cfg2.log 2>&1; echo "sq rc=$?"
bash tools/collect_profiles.sh idx1B --index-kmers 1000000000 --reads 28000000 > gpurun_out/collect_1B.log 2>&1; echo "1B profiles rc=$?"; tail -3 gpurun_out/collect_1B.log
du -sh gpurun_out/prof_cfg2 gpurun_out/prof_idx1B gpurun_out/sq_cfg2 2>/dev/null
