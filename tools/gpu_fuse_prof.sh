#!/bin/bash
# kernel-level comparison of the sampler with and without the in-launch self-attention (rocprofv3 kernel stats of the same command)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for f in 0 1; do
  export TCDIFF_FUSE_SA=$f
  rm -rf gpurun_out/prof_f$f
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f$f -o fuse$f -- python3 bench.py --steps 60 --warmup 5 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs > gpurun_out/prof_f$f.json 2> gpurun_out/prof_f$f.err
  tail -1 gpurun_out/prof_f$f.json | cut -c1-200
  find gpurun_out/prof_f$f -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} | cut -c1-150
  find gpurun_out/prof_f$f -name "*kernel_trace.csv" -delete
done
