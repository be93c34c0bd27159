#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_chain_gpu.py -q -x -s -m gpu -k "front" 2>&1 | tail -8
python -m pytest tests/test_parity_gpu.py tests/test_chain_gpu.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2; do
for w in 0 1; do
  TCDIFF_FRONT=$w python bench.py --steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline 2>gpurun_out/ab_err.log > gpurun_out/ab_w$w.json
  echo -n "FRONT=$w: "; python tools/show_bench.py gpurun_out/ab_w$w.json
done; done
