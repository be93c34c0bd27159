#!/bin/bash
# round 4, training step through the row-block GEMM: parity of the step, then the step time with and without it (same box)
python -m pytest tests/test_train_step_gpu.py tests/test_train_gpu.py tests/test_train_dist_gpu.py -m gpu -x -q 2>&1 | tail -5
for rows in 1 0 1 0; do
  for b in 32 4; do echo "TCDIFF_TRAIN_ROWS=$rows batch $b: $(TCDIFF_TRAIN_ROWS=$rows python tools/train_bench.py --batch $b --iters 10 2>/dev/null | tail -1 | cut -c1-220)"; done
done | tee gpurun_out/r04_train_rows_ab.txt
