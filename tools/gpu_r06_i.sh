#!/bin/bash
# round 6, visit I: the small-job form of the layer (chain_split.hip): kernel tests, whole-network tests, small-job timings
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_split_gpu.py -q -m gpu -s 2>&1 | grep -E "L=|passed|failed|Error|assert|rror" | cut -c1-220 | tee gpurun_out/r06_chain_split_tests.log
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_chain_selfatt_gpu.py -q -m gpu -x -k "c1 or network or ddim or several" 2>&1 | tail -5
timeout 900 python tools/small_batch.py 2 2s 2>&1 | tail -3 | tee gpurun_out/r06_small_batch_split.txt
