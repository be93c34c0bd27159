#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ov
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --ddpm-steps 200 > gpurun_out/prof_ov.log 2>&1; echo "trace rc=$?"
python3 tools/trace_overlap.py gpurun_out/prof_ov | tee gpurun_out/overlap.txt
rm -rf gpurun_out/prof_ov
