#!/bin/bash
# round 5, visit n: the in-kernel cross-attention as ONE pass over the K / V tiles (four row tiles per tile) -- tests, launch A/B, sampler A/B
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TCDIFF_LIB_PATH=tools/probe/libtc_ONEPASS.so timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do for v in BASE ONEPASS; do
  echo "== $v"; TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_full_bench.py --forms 8 --blocks 1,225 --reps 3 2>&1 | grep "waves:"
done; done | tee gpurun_out/r05_xatt_onepass.txt
F="--steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3; do for v in BASE ONEPASS; do
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 600 python bench.py $F 2>gpurun_out/ab_err.log > gpurun_out/ab_$v.json
  echo -n "$v: "; python tools/show_bench.py gpurun_out/ab_$v.json
done; done 2>&1 | tee -a gpurun_out/r05_xatt_onepass.txt
