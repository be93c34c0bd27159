#!/bin/bash
# new ADVICE tests, chain tools with 64-row blocks pinned, 1000-step parity lines
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_step_gpu.py -x -q -m gpu -k "stale or reallocated" 2>&1 | tail -4
TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_chain_stamps.txt; grep -E "fused layer|last wave|shader clock" gpurun_out/r04_chain_stamps.txt
timeout 300 python tools/chain_bench.py 2>/dev/null | grep "chain" > gpurun_out/r04_chain_block_scaling.txt; cat gpurun_out/r04_chain_block_scaling.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -s -q -k "1000" 2>&1 | grep -E "1000|passed|failed" | tail -6
