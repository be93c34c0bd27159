#!/bin/bash
# round 6, visit O: two weight rings per wave in the split kernels
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_split_gpu.py -q -m gpu -s 2>&1 | grep -E "L=|vs fused|passed|failed|Error|assert|rror" | cut -c1-220 > gpurun_out/r06_chain_split_tests.log; grep -E "passed|failed|rror" gpurun_out/r06_chain_split_tests.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "c1 or ddim" 2>&1 | tail -3
TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so timeout 300 python tools/split_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_split_stamps.txt
timeout 900 python tools/small_batch.py 2 2>&1 | tail -2 | tee gpurun_out/r06_small_batch_split.txt
