#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-kernel-profile --ddpm-steps 200"
for mode in dual single; do
  if [ $mode = single ]; then export TCDIFF_DUAL=0; else export TCDIFF_DUAL=1; fi
  rm -rf gpurun_out/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$mode -- python3 $ARGS > gpurun_out/prof_$mode.log 2>&1; echo "$mode rc=$?"
  grep -o '"value": [0-9.]*' gpurun_out/prof_$mode.log | head -1
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2d_kernel_stats_$mode.csv
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-60s calls %6s avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  find gpurun_out/prof_$mode -name "*kernel_trace.csv" -delete
done
unset TCDIFF_DUAL
for d in 0 1; do TCDIFF_DUAL=$d timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-kernel-profile > gpurun_out/r2d_bench_dual$d.log 2>&1; echo "dual=$d: $(grep -o '"value": [0-9.]*' gpurun_out/r2d_bench_dual$d.log | head -1)"; done
