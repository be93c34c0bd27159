#!/bin/bash
# round 6, visit H: x stores behind the epilogue loop (vmcnt is one in-order counter) and several steps per captured graph
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_chain_selfatt_gpu.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do
  echo -n "stores late (default): "; timeout 200 python tools/chain_sa_bench.py --reps 3 --only "self-attention" 2>&1 | grep "rows,"
  echo -n "stores early (r2-r5) : "; TCDIFF_LIB_PATH=tools/probe/libtc_STEARLY.so timeout 200 python tools/chain_sa_bench.py --reps 3 --only "self-attention" 2>&1 | grep "rows,"
done | tee gpurun_out/r06_store_order_launch.txt
F="--steps 12 --warmup 2 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3; do
  echo "stores late,  20 steps/graph: $(timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c80-120)"
  echo "stores early, 20 steps/graph: $(TCDIFF_LIB_PATH=tools/probe/libtc_STEARLY.so timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c80-120)"
  echo "stores late,   1 step/graph : $(TCDIFF_GRAPH_STEPS=1 timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c80-120)"
done | tee gpurun_out/r06_store_order_graph_steps_ab.txt
