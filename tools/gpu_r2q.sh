#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_chain_gpu.py tests/test_parity_gpu.py -q -x -m gpu 2>&1 | tail -2
bash tools/gpu_abtrace.sh "$@" 2>&1 | grep "==\|chain_kernel<3\|chain_kernel<4"
bash tools/ab_run.sh "$@"; bash tools/ab_run.sh "$@"
