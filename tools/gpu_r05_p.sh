#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "bf16x3" 2>&1 | grep -E "bf16x3|passed|failed|Error|error" | tail -16 | tee gpurun_out/r05_bf16x3_parity.log
F="--steps 1 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs --ddpm-steps 200 --dtype bf16x3"
for rep in 1 2; do for v in 1 0; do echo -n "TCDIFF_X3_ROWLN=$v: "; TCDIFF_X3_ROWLN=$v timeout 900 python bench.py $F 2>gpurun_out/x3_err.log > gpurun_out/x3_r$v.json; python tools/show_bench.py gpurun_out/x3_r$v.json | head -1; done; done | tee gpurun_out/r05_bf16x3_rowln_ab.txt
tail -2 gpurun_out/x3_err.log
