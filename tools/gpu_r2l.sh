#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_chain_gpu.py tests/test_parity_gpu.py -q -x -m gpu 2>&1 | tail -3
bash tools/ab_run.sh "$@"
