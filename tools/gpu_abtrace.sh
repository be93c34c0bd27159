#!/bin/bash
# same-box rocprofv3 kernel stats (100 + 100 DDPM steps) for each tools/probe/libtc_<name>.so named on the command line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/prof_ab
  export TCDIFF_LIB_PATH=tools/probe/libtc_$v.so
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --ddpm-steps 100 --no-cpu-baseline --no-kernel-profile --no-parity-mode > gpurun_out/prof_ab.log 2>&1
  f=$(find gpurun_out/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -7 "$f" | cut -c1-70,100-150
  rm -rf gpurun_out/prof_ab
done
