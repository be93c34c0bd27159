#!/bin/bash
mkdir -p gpurun_out
python -m tcdiff_amd.build > gpurun_out/build.log 2>&1 || { tail gpurun_out/build.log; exit 1; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "rowln" 2>&1 | tail -5
timeout 600 python bench.py --steps 1 --warmup 1 --ddpm-steps 100 --cpu-seconds 1 > gpurun_out/bench_short.log 2>&1; python tools/show_bench.py gpurun_out/bench_short.log 2>/dev/null | head -30
TCDIFF_EXTRA_HIPCC_FLAGS="-DTC_STAMP" python -m tcdiff_amd.build --force > gpurun_out/stamp_build.log 2>&1 || { tail gpurun_out/stamp_build.log; exit 1; }
timeout 300 python tools/microbench3.py 2>&1 | grep rowln > gpurun_out/stamp3.log
cat gpurun_out/stamp3.log
