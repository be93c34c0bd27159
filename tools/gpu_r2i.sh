#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_chain_gpu.py -q -x -m gpu 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline 2>gpurun_out/bench_err.log > gpurun_out/bench_r2i.json; python tools/show_bench.py gpurun_out/bench_r2i.json
python tools/chain_stamps.py 2>&1 | tail -46
