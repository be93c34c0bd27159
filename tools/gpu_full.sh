#!/bin/bash
# Full-size bench (driver contract defaults) + rocprofv3 kernel stats (200 DDPM steps) + HBM PMC passes (30 steps:
# longer PMC runs have crashed rocprofv3) of the same command.  Outputs in gpurun_out/; copy the summaries to profiles/.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench_full.log
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --ddpm-steps"
rm -rf gpurun_out/prof_full gpurun_out/prof_fetch gpurun_out/prof_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -- python3 $ARGS 200 > gpurun_out/prof_full.log 2>&1; echo "trace rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 $ARGS 30 > gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 $ARGS 30 > gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
python3 tools/pmc_summary.py gpurun_out/prof_fetch > gpurun_out/pmc_FETCH_SIZE_summary.txt; python3 tools/pmc_summary.py gpurun_out/prof_write > gpurun_out/pmc_WRITE_SIZE_summary.txt
python3 tools/make_pmc_json.py gpurun_out/pmc_FETCH_SIZE_summary.txt gpurun_out/pmc_WRITE_SIZE_summary.txt gpurun_out/pmc.json | head -30
f=$(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_full.csv; head -12 "$f" | cut -c1-200
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
