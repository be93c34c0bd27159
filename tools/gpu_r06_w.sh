#!/bin/bash
# round 6, visit W: the duplicated 16-stage products (linear3, the merged form's fc) with rotated k-quarters
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_split_gpu.py -q -m gpu 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "c1 or ddim" 2>&1 | tail -2
TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so timeout 300 python tools/split_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_split_stamps_rotated.txt; grep -A11 "part 4" gpurun_out/r06_split_stamps_rotated.txt
timeout 900 python tools/small_batch.py 2 2>&1 | tail -1 | tee gpurun_out/r06_small_batch_split.txt
