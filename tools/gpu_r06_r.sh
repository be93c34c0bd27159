#!/bin/bash
# round 6, visit R: the tests visit Q failed (small-M kernel now opt-in), the new tests (part 0, use_rotary=False training), small jobs
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests/test_bench_dist_gpu.py tests/test_train_kernels_gpu.py -q -m gpu -x -k "two_ranks or fused_activation" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_train_step_gpu.py -q -m gpu -s -k "row_block or without_rotary" 2>&1 | grep -E "^\[|\.\[|passed|failed|Error|assert" | cut -c1-260 | tee gpurun_out/r06_train_no_rotary.log
timeout 900 python -m pytest tests/test_chain_split_gpu.py tests/test_kernels_gpu.py -q -m gpu -s -k "front_part or small_products or small_job" 2>&1 | grep -E "part 0|vs fused|passed|failed|Error|assert" | cut -c1-220
timeout 900 python tools/small_batch.py 2 2>&1 | tail -2 | tee gpurun_out/r06_small_batch_split.txt
