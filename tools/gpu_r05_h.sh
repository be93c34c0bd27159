#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step --no-other-configs --ddpm-steps 40 --dtype bf16x3"
rm -rf gpurun_out/prof_x3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x3 -- python3 $ARGS > gpurun_out/prof_x3.log 2>&1
f=$(find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_kernel_stats_bf16x3_40steps.csv; head -24 "$f" | cut -c1-150
rm -rf gpurun_out/prof_x3
