#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "bf16x3" 2>&1 | grep -E "bf16x3|passed|failed|Error|error" | tail -20 | tee gpurun_out/r05_bf16x3_parity.log
F="--steps 1 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs --ddpm-steps 200"
for dt in f32 bf16x3; do echo -n "$dt: "; timeout 900 python bench.py $F --dtype $dt 2>gpurun_out/x3_err.log > gpurun_out/x3_$dt.json; python tools/show_bench.py gpurun_out/x3_$dt.json | head -1; done | tee gpurun_out/r05_bf16x3_speed.txt
tail -3 gpurun_out/x3_err.log
timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -4
