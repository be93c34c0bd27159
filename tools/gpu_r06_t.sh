#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "c1 or ddim or prologue or c2_ddpm" 2>&1 | tail -3
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step --no-other-configs --no-pmc"
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 $ARGS --ddpm-steps 100 > gpurun_out/prof_trace.log 2>&1
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); grep -E "prologue|sampler_update" "$f" | cut -c1-140
rm -rf gpurun_out/prof_trace
