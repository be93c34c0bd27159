#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_gpu_suite_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke" | tee gpurun_out/r05_smoke.log
