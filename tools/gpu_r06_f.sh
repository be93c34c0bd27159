#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests/test_train_step_gpu.py -q -m gpu -s -k "two_training_steps or two_roundings" 2>&1 | grep -E "\[bf16|passed|failed|Error|assert" | cut -c1-600 | tee gpurun_out/r06_train_step_vs_draws.log
