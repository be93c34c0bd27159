#!/bin/bash
# inter-kernel gaps of the replayed sampler step (rocprofv3 kernel trace timestamps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gap
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step --no-other-configs --ddpm-steps 100 > gpurun_out/prof_gap.log 2>&1
f=$(find gpurun_out/prof_gap -name "*kernel_trace.csv" | head -1)
python tools/gap_analysis.py "$f" 0.6 | tee gpurun_out/r04_gap_analysis.txt
rm -rf gpurun_out/prof_gap
