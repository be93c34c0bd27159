#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "prologue or counter or sampler" 2>&1 | tail -5
python -m pytest tests/test_parity_gpu.py -q -x -m gpu 2>&1 | tail -5
python bench.py --steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline 2>gpurun_out/bench_err.log > gpurun_out/bench_r2f.json; python tools/show_bench.py gpurun_out/bench_r2f.json
