#!/bin/bash
# round 6, visit S: parts 1 + 2 of the small-job layer as one launch
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_chain_split_gpu.py -q -m gpu -s 2>&1 | grep -E "L=|vs fused|part 0|passed|failed|Error|assert|rror" | cut -c1-220 > gpurun_out/r06_chain_split_tests.log; grep -E "merged.*vs fused|passed|failed|rror" gpurun_out/r06_chain_split_tests.log | head -20
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "c1 or ddim" 2>&1 | tail -3
timeout 900 python tools/small_batch.py 2m 2n 2>&1 | tail -2 | tee gpurun_out/r06_small_batch_merge.txt
