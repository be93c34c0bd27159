#!/bin/bash
# round 5, visit f: split-bf16 mode parity + speed; (appended) MFMA rate probe and the attention variants of visit e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./tools/probe/mfma_rate_probe | tee gpurun_out/r05_mfma_rate_probe.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "bf16x3" 2>&1 | grep -E "bf16x3|passed|failed|Error|error" | tail -20 | tee gpurun_out/r05_bf16x3_parity.log
F="--steps 1 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs --ddpm-steps 200"
for dt in f32 bf16x3; do echo -n "$dt: "; timeout 900 python bench.py $F --dtype $dt 2>gpurun_out/x3_err.log | python tools/show_bench.py /dev/stdin; done | tee gpurun_out/r05_bf16x3_speed.txt
bash tools/gpu_r05_e.sh > /dev/null 2>&1; cat gpurun_out/r05_attn_variants.txt
