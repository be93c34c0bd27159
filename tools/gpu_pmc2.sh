#!/bin/bash
# SQ-level counters of a 20-step bench run, one rocprofv3 pass per counter set ($SETS: space-separated, commas inside a set)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repo root)}"
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|TCP|TA|TCC|GRBM)_[A-Z0-9_]+" | sort -u > gpurun_out/counters_avail.txt; wc -l gpurun_out/counters_avail.txt
ARGS="bench.py --steps 1 --warmup 1 --ddpm-steps 20 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step"
i=0
: > gpurun_out/pmc2_summary.txt
for set in $SETS; do
  i=$((i+1)); rm -rf gpurun_out/prof_s$i
  timeout 600 rocprofv3 --kernel-trace --pmc ${set//,/ } --output-format csv -d gpurun_out/prof_s$i -- python3 $ARGS > gpurun_out/prof_s$i.log 2>&1; echo "set $i ($set) rc=$?"
  echo "== $set" >> gpurun_out/pmc2_summary.txt
  python3 tools/pmc_summary.py gpurun_out/prof_s$i | head -12 >> gpurun_out/pmc2_summary.txt
  rm -rf gpurun_out/prof_s$i
done
cat gpurun_out/pmc2_summary.txt | cut -c1-400
