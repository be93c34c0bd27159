#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="--steps 1 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs --ddpm-steps 200 --dtype bf16x3"
for rep in 1 2; do for d in 0 1; do echo -n "TCDIFF_X3_DEEP=$d: "; TCDIFF_X3_DEEP=$d timeout 900 python bench.py $F 2>gpurun_out/x3_err.log > gpurun_out/x3_d$d.json; python tools/show_bench.py gpurun_out/x3_d$d.json | head -1; done; done
