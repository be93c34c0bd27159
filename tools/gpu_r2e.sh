#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2e_all.log 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r2e_all.log
for d in 1 0; do TCDIFF_DUAL=$d timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-kernel-profile > gpurun_out/r2e_bench_dual$d.log 2>&1; echo "dual=$d: $(grep -o '"value": [0-9.]*' gpurun_out/r2e_bench_dual$d.log | head -1)"; done
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-kernel-profile --ddpm-steps 200"
export TCDIFF_DUAL=0
rm -rf gpurun_out/prof_e
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -- python3 $ARGS > gpurun_out/prof_e.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/prof_e -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2e_kernel_stats_single.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-60s calls %6s avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
find gpurun_out/prof_e -name "*kernel_trace.csv" -delete
