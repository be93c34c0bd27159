#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -30 > gpurun_out/gemm_tests.log
tail -15 gpurun_out/gemm_tests.log
TCDIFF_GEMM_KERNEL=1 timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1
TCDIFF_GEMM_KERNEL=2 timeout 300 python tools/gemm_bench.py >> gpurun_out/gemm_bench.log 2>&1
grep kernel= gpurun_out/gemm_bench.log
