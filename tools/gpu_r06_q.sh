#!/bin/bash
# round 6, interim check of the committed tree: whole GPU suite, smoke, default bench
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06_gpu_suite_final.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r06_smoke.log
python bench.py 2>gpurun_out/bench_default_err.log > gpurun_out/r06_bench_full_1000steps.json; tail -c 1500 gpurun_out/r06_bench_full_1000steps.json
