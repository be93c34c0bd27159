#!/bin/bash
# One GPU-box visit: kernel unit tests, end-to-end parity, smoke, a short bench.  Logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/box.txt; nproc >> gpurun_out/box.txt; lscpu | grep "Model name" >> gpurun_out/box.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rA --no-header -p no:cacheprovider 2>&1 | tail -400 > gpurun_out/kernels.log
echo "kernels rc=$?" ; tail -40 gpurun_out/kernels.log
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -rA -s --no-header -p no:cacheprovider 2>&1 | tail -400 > gpurun_out/parity.log
echo "parity rc=$?"; tail -60 gpurun_out/parity.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 900 python bench.py --steps 1 --warmup 1 --ddpm-steps ${DDPM_STEPS:-100} --cpu-seconds 5 > gpurun_out/bench_short.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench_short.log
