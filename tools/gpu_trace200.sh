#!/bin/bash
# rocprofv3 kernel stats of 200 (+200 warm-up) DDPM steps of the bench workload -> gpurun_out/r02_kernel_stats.csv
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 1 --warmup 1 --ddpm-steps 200 --no-cpu-baseline --no-kernel-profile --no-parity-mode > gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_kernel_stats.csv; head -${ROWS:-14} "$f" | cut -c1-60,100-200
find gpurun_out/prof_trace -name "*kernel_trace.csv" -delete
