#!/bin/bash
mkdir -p gpurun_out
TCDIFF_EXTRA_HIPCC_FLAGS="-DTC_STAMP" python -m tcdiff_amd.build --force > gpurun_out/stamp_build.log 2>&1 || { tail gpurun_out/stamp_build.log; exit 1; }
TCDIFF_GEMM_KERNEL=1 timeout 300 python tools/microbench3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stamp3.log
cat gpurun_out/stamp3.log
