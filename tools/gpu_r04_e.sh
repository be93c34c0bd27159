#!/bin/bash
# same-box A/B: round-3 tree (tools/probe/r3_tree, 32x32x16 chain kernel) against the working tree; then the new tests and the
# small-batch modes
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3; do
  (cd tools/probe/r3_tree && python bench.py $B 2>/dev/null | python ../../show_bench.py /dev/stdin | sed 's/^/r3:  /')
  python bench.py $B 2>/dev/null | python tools/show_bench.py /dev/stdin | sed 's/^/r4:  /'
done
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_train_dist_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -s -k "attribution" 2>&1 | grep -E "guided evaluation|bf16|f32 parity|passed|failed" 
timeout 1200 python tools/small_batch.py 2>&1 | tail -4
