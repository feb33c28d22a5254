#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box:
#   tools/collect_profiles.sh <tag> <bench args...>      e.g.  tools/collect_profiles.sh cfg2 --config 2
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,rdsize,wrsize}/ (CSV) — copy the summaries into profiles/.
# Counters are collected in passes of their own, with --kernel-trace only (never with a runtime / sys trace).
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
common="--steps 4 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-records-host-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $common "$@" > $out/stats.json 2> $out/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py $common "$@" > /dev/null 2> $out/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py $common "$@" > /dev/null 2> $out/write.err || exit 1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $out/rdsize -- python3 bench.py $common "$@" > /dev/null 2> $out/rdsize.err || exit 1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/wrsize -- python3 bench.py $common "$@" > /dev/null 2> $out/wrsize.err || exit 1
# keep only the CSVs that are read afterwards (the merge back is capped at 64 MiB)
find $out -name "*_agent_info.csv" -delete
find $out -name "*kernel_trace.csv" -size +8M -delete
ls $out/stats/*/ | head
