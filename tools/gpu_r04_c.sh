#!/bin/bash
# new 16x16x32 chain kernel: unit tests, goldens, stamps, quick bench
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_dist_gpu.py -x -q -m gpu 2>&1 | tail -8
TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stamps_mfma16.txt
grep -E "fused layer|last wave|shader clock" gpurun_out/r04_stamps_mfma16.txt
timeout 300 python tools/chain_bench.py 2>/dev/null | grep "chain" > gpurun_out/r04_chain_block_scaling_mfma16.txt; cat gpurun_out/r04_chain_block_scaling_mfma16.txt
python bench.py --steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs 2>gpurun_out/ab_err.log > gpurun_out/ab_mfma16.json; python tools/show_bench.py gpurun_out/ab_mfma16.json
