#!/bin/bash
# kernel unit tests selected by $1 (pytest -k expression) + the short bench with its per-kernel table
mkdir -p gpurun_out
python -m tcdiff_amd.build > gpurun_out/build.log 2>&1 || { tail gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "$1" 2>&1 | tail -4
timeout 600 python bench.py --steps 1 --warmup 1 --ddpm-steps 200 --cpu-seconds 1 > gpurun_out/bench_short.log 2>&1; python tools/show_bench.py gpurun_out/bench_short.log 2>/dev/null | head -12
