#!/bin/bash
# Same-box A/B harness of the round-5 experiments (one gpurun visit each; what they measured is in profiles/r05_*.txt).
#   bash tools/gpu_experiments.sh wave_forms      -- chain kernel, 8-wave vs 4-wave form: tests, fused-layer launch, sampler
#   bash tools/gpu_experiments.sh stamps          -- in-kernel stamps of the fused layer (both forms) incl. per-stage stamps of one GEMM phase
#   bash tools/gpu_experiments.sh x3              -- split-bf16 mode: parity tests, f32 vs bf16x3 speed, per-kernel profile
#   bash tools/gpu_experiments.sh libs A B ..     -- fused-layer launch + sampler for tools/probe/libtc_<NAME>.so variants (tools/ab_build.sh)
#   bash tools/gpu_experiments.sh attn A B ..     -- self-attention launch for library variants
#   bash tools/gpu_experiments.sh probe           -- the co-issue and MFMA-rate probes
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repo root)}"
F="--steps 2 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
what=$1; shift
case $what in
wave_forms)
  timeout 1500 python -m pytest tests/test_chain_gpu.py -q -m gpu -x 2>&1 | tail -4
  timeout 600 python tools/chain_full_bench.py 2>&1 | grep -v amdgpu.ids
  for rep in 1 2; do for nw in 8 4; do
    TCDIFF_CHAIN_NW=$nw timeout 600 python bench.py $F 2>gpurun_out/ab_err.log > gpurun_out/ab_nw$nw.json
    echo -n "nw=$nw: "; python tools/show_bench.py gpurun_out/ab_nw$nw.json
  done; done ;;
stamps)
  for nw in 8 4; do NW=$nw TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/chain_stamps_nw$nw.txt; done
  paste -d'|' <(cut -c1-62 gpurun_out/chain_stamps_nw8.txt) <(cut -c52-62 gpurun_out/chain_stamps_nw4.txt) | head -90 ;;
x3)
  timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "bf16x3" 2>&1 | grep -E "bf16x3|passed|failed|rror" | tail -16
  for dt in f32 bf16x3; do echo -n "$dt: "; timeout 900 python bench.py $F --steps 1 --ddpm-steps 200 --dtype $dt 2>gpurun_out/x3_err.log > gpurun_out/x3_$dt.json; python tools/show_bench.py gpurun_out/x3_$dt.json | head -1; done
  rm -rf gpurun_out/prof_x3
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x3 -- python3 bench.py $F --steps 1 --ddpm-steps 40 --dtype bf16x3 > gpurun_out/prof_x3.log 2>&1
  f=$(find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_bf16x3_40steps.csv; head -8 "$f" | cut -c1-150; rm -rf gpurun_out/prof_x3 ;;
libs)
  for rep in 1 2; do for v in "$@"; do
    echo "== $v"; TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_full_bench.py --forms 8 --blocks 1,225 --reps 3 2>&1 | grep "waves:"
  done; done
  for rep in 1 2 3; do for v in "$@"; do
    TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 600 python bench.py $F 2>gpurun_out/ab_err.log > gpurun_out/ab_$v.json
    echo -n "$v: "; python tools/show_bench.py gpurun_out/ab_$v.json
  done; done ;;
attn)
  for rep in 1 2 3; do for v in "$@"; do
    TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 120 python tools/attn_infer_bench.py 32 16 2>&1 | grep "workgroups"
  done; done ;;
probe)
  timeout 300 ./tools/probe/coissue_probe | tail -40; ./tools/probe/mfma_rate_probe ;;
*) echo "unknown experiment $what"; exit 2 ;;
esac
