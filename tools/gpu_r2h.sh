#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py -q -x -m gpu 2>&1 | tail -3
python tools/train_bench.py --batch 4 2>&1 | tail -1 | tee gpurun_out/train_bench_b4.json
python tools/train_bench.py --batch 32 2>&1 | tail -1 | tee gpurun_out/train_bench_b32.json
