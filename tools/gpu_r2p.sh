#!/bin/bash
mkdir -p gpurun_out
TCDIFF_LIB_PATH=tools/probe/libtc_XP3.so python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "c2 or bf16" 2>&1 | tail -2
TCDIFF_LIB_PATH=tools/probe/libtc_XP2.so python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "c2 or bf16" 2>&1 | tail -2
bash tools/ab_run.sh "$@"
