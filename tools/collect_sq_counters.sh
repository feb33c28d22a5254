#!/bin/bash
# SQ counters of the radix passes (two rocprofv3 --pmc passes of their own, --kernel-trace only), summarised per kernel:
#   tools/collect_sq_counters.sh <tag> [bench args...]  ->  gpurun_out/sq_<tag>.json
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
rm -rf "$out"; mkdir -p "$out"
common="--steps 2 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-records-host-leg"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $out/a -- python3 bench.py $common "$@" > /dev/null 2> $out/a.err || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d $out/b -- python3 bench.py $common "$@" > /dev/null 2> $out/b.err || exit 1
python3 tools/pmc_summary.py $out k_rx_p1 k_rx_p2 k_rx_p3 > gpurun_out/sq_$tag.json
find $out -name "*kernel_trace.csv" -delete
find $out -name "*_agent_info.csv" -delete
head -c 400 gpurun_out/sq_$tag.json
