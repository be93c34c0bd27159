#!/bin/bash
# bench every tools/probe/libtc_<name>.so named on the command line, twice, interleaved (same box)
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    TCDIFF_LIB_PATH=tools/probe/libtc_$v.so python bench.py --steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs 2>gpurun_out/ab_err.log > gpurun_out/ab_$v.json
    echo -n "$v: "; python tools/show_bench.py gpurun_out/ab_$v.json
  done
done
