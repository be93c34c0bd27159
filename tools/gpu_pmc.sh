#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 1 --ddpm-steps 30 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 $ARGS > gpurun_out/prof_$c.log 2>&1; echo "$c rc=$?"
  python3 tools/pmc_summary.py gpurun_out/prof_$c > gpurun_out/pmc_${c}_summary.txt; head -6 gpurun_out/pmc_${c}_summary.txt
done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
