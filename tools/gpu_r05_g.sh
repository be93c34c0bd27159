#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="--steps 1 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs --ddpm-steps 200"
for dt in f32 bf16x3; do echo -n "$dt: "; timeout 900 python bench.py $F --dtype $dt 2>gpurun_out/x3_err.log > gpurun_out/x3_$dt.json; python tools/show_bench.py gpurun_out/x3_$dt.json | head -1; done | tee gpurun_out/r05_bf16x3_speed.txt
for rep in 1 2 3; do for v in BASE ATT_P ATT_UP; do
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 120 python tools/attn_infer_bench.py 32 2>&1 | grep "workgroups"
done; done | tee gpurun_out/r05_attn_variants_b32.txt
