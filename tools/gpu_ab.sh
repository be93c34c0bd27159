#!/bin/bash
# A/B of environment switches on the short bench: usage  gpu_ab.sh "VAR=a VAR=b ..."  (each token one run, twice)
mkdir -p gpurun_out
: > gpurun_out/ab.log
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout 600 python bench.py --steps 1 --warmup 1 --ddpm-steps 200 --no-cpu-baseline > gpurun_out/ab_one.log 2>&1
  echo "$cfg: $(python tools/show_bench.py gpurun_out/ab_one.log | head -1)" >> gpurun_out/ab.log
done; done
cat gpurun_out/ab.log
