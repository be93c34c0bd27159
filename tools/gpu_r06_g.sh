#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests/test_train_kernels_gpu.py -q -m gpu -s -k "attention" 2>&1 | grep -E "L=|passed|failed|Error|assert" | cut -c1-300
timeout 1500 python -m pytest tests/test_train_step_gpu.py tests/test_train_gpu.py -q -m gpu -s -x 2>&1 | grep -E "\[bf16\] step|error SHAPE|derived bound|passed|failed|Error|assert" | cut -c1-420 | tee gpurun_out/r06_train_step_vs_draws.log
for rep in 1 2 3; do for f in 0 1; do
  echo "TCDIFF_TRAIN_OLO=$f: $(TCDIFF_TRAIN_OLO=$f python tools/train_bench.py --batch 32 --iters 10 2>/dev/null | tail -1 | cut -c1-150)"
done; done | tee gpurun_out/r06_train_olo_ab.txt
