#!/bin/bash
# round-2 measurement pass: full GPU suite, default bench, rocprofv3 kernel stats (200 steps), FETCH/WRITE PMC passes (30 steps)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -x -m gpu 2>&1 | tail -4
python -m pytest tests/test_parity_gpu.py -m gpu -s -q -k "c2 or bf16 or drift" > gpurun_out/r02_parity_at_benchmarked_config.log 2>&1; tail -3 gpurun_out/r02_parity_at_benchmarked_config.log
python bench.py 2>gpurun_out/bench_default_err.log > gpurun_out/bench_default.json; tail -c 3000 gpurun_out/bench_default.json
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py --steps 1 --warmup 1 --ddpm-steps 200 --no-cpu-baseline --no-kernel-profile --no-parity-mode > gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_kernel_stats.csv; head -24 "$f" | cut -c1-160
find gpurun_out/prof_trace -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --steps 1 --warmup 1 --ddpm-steps 30 --no-cpu-baseline --no-kernel-profile --no-parity-mode > gpurun_out/prof_$c.log 2>&1
  echo "$c rc=$?"
  python3 tools/pmc_summary.py gpurun_out/prof_$c > gpurun_out/r02_pmc_${c}_30steps.txt
  rm -rf gpurun_out/prof_$c
done
python3 tools/make_pmc_json.py gpurun_out/r02_pmc_FETCH_SIZE_30steps.txt gpurun_out/r02_pmc_WRITE_SIZE_30steps.txt gpurun_out/r02_pmc.json 0
