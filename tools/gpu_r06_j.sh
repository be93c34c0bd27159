#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_small
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python3 tools/small_job.py > gpurun_out/prof_small.log 2>&1
tail -3 gpurun_out/prof_small.log
f=$(find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_kernel_stats_small_job_ddim50_1clip.csv; head -16 "$f" | cut -c1-150
rm -rf gpurun_out/prof_small
