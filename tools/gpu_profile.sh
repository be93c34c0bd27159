#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC passes) of a short bench run; summaries land in gpurun_out/prof_*.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 1 --ddpm-steps ${DDPM_STEPS:-30} --no-cpu-baseline"
rm -rf gpurun_out/prof_trace gpurun_out/prof_pmc*
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 $ARGS > gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"; tail -2 gpurun_out/prof_trace.log | cut -c1-400
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); echo $f; head -30 "$f"
if [ -n "$PMC" ]; then
  i=0
  for set in $PMC; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc ${set//,/ } --output-format csv -d gpurun_out/prof_pmc$i -- python3 $ARGS > gpurun_out/prof_pmc$i.log 2>&1
    echo "pmc$i ($set) rc=$?"
    python3 tools/pmc_summary.py gpurun_out/prof_pmc$i | head -40
  done
fi
# keep only the small summaries (traces are large)
find gpurun_out/prof_trace -name "*kernel_trace.csv" -size +20M -delete
