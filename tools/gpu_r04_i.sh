#!/bin/bash
# round 4, weight gradients on a side stream: parity of the step (1 and 2 ranks), then the step time with and without (same box)
python -m pytest tests/test_train_step_gpu.py tests/test_train_gpu.py tests/test_train_dist_gpu.py -m gpu -q 2>&1 | tail -5
for ws in 1 0 1 0; do
  for b in 32; do echo "TCDIFF_TRAIN_WGRAD_STREAM=$ws batch $b: $(TCDIFF_TRAIN_WGRAD_STREAM=$ws python tools/train_bench.py --batch $b --iters 10 2>/dev/null | tail -1 | cut -c1-160)"; done
done | tee gpurun_out/r04_train_wgrad_stream_ab.txt
