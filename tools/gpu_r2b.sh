#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q -s > gpurun_out/r2b_chain.log 2>&1; echo "chain tests rc=$?"
grep -E "chain|passed|failed|Error|error|assert" gpurun_out/r2b_chain.log | tail -30
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b_alltests.log 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r2b_alltests.log
timeout 600 python bench.py --no-cpu-baseline --no-parity-mode > gpurun_out/r2b_bench.log 2>&1; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/r2b_bench.log 2>/dev/null | head -40 || tail -c 2500 gpurun_out/r2b_bench.log
TCDIFF_CHAIN=0 timeout 600 python bench.py --no-cpu-baseline --no-parity-mode --no-kernel-profile > gpurun_out/r2b_bench_nochain.log 2>&1; echo "bench(no chain) rc=$?"; grep -o '"value": [0-9.]*' gpurun_out/r2b_bench_nochain.log | head -1
