#!/bin/bash
# per-job fixed cost of p_sample_loop: time of jobs with 100 / 300 / 1000 DDPM steps (same schedule length T=1000? no: T = steps)
for T in 100 300 1000; do
  python bench.py --steps 3 --warmup 1 --ddpm-steps $T --no-kernel-profile --no-parity-mode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('T=$T: job', r['ms_per_step'], 'ms')"
done
