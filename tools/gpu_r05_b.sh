#!/bin/bash
# round 5, visit b: the 4-wave chain form -- tests, fused-layer launch A/B, sampler A/B (same box, interleaved)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_chain_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r05_chain_tests.log; tail -4 gpurun_out/r05_chain_tests.log
timeout 600 python tools/chain_full_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_chain_full_bench.txt; cat gpurun_out/r05_chain_full_bench.txt
F="--steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2; do for nw in 8 4; do
  TCDIFF_CHAIN_NW=$nw timeout 600 python bench.py $F 2>gpurun_out/ab_err.log > gpurun_out/ab_nw$nw.json
  echo -n "nw=$nw: "; python tools/show_bench.py gpurun_out/ab_nw$nw.json
done; done 2>&1 | tee gpurun_out/r05_nw_ab.txt
