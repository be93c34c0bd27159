#!/bin/bash
# tests of the fold + MT build, then a longer same-box A/B against the round-3 tree (5 pairs)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_chain_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -4
B="--steps 3 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3 4 5; do
  (cd tools/probe/r3_tree && python bench.py $B 2>/dev/null | python ../../show_bench.py /dev/stdin | sed 's/^/r3:  /')
  python bench.py $B 2>/dev/null | python tools/show_bench.py /dev/stdin | sed 's/^/r4:  /'
done
