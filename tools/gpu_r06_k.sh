#!/bin/bash
# round 6, visit K: split kernels with every independent load issued at the top: kernel tests, small-job timings, per-kernel durations
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_split_gpu.py -q -m gpu -s 2>&1 | grep -E "L=|passed|failed|Error|assert|rror" | cut -c1-220 | tee gpurun_out/r06_chain_split_tests.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "c1 or ddim" 2>&1 | tail -3
timeout 900 python tools/small_batch.py 2 2s 2>&1 | tail -3 | tee gpurun_out/r06_small_batch_split.txt
rm -rf gpurun_out/prof_small
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python3 tools/small_job.py > gpurun_out/prof_small.log 2>&1
tail -3 gpurun_out/prof_small.log
f=$(find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_kernel_stats_small_job_ddim50_1clip.csv; head -16 "$f" | cut -c1-150
rm -rf gpurun_out/prof_small
