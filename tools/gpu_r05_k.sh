#!/bin/bash
# round 5, visit k: the whole GPU suite + the default bench (banking pass)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05_gpu_suite.log
timeout 900 python -m pytest tests/test_train_step_gpu.py -m gpu -s -q -k "rounds_where" 2>&1 | grep -E "\[bf16|passed|failed" | tee gpurun_out/r05_train_step_vs_emulation.log
timeout 900 python bench.py 2>gpurun_out/bench_default_err.log > gpurun_out/r05_bench_mid.json; python tools/show_bench.py gpurun_out/r05_bench_mid.json
