#!/bin/bash
# measurement pass of a round (one box visit; R=rNN names the outputs): full GPU suite, parity log at the benchmarked config, default bench, rocprofv3 kernel
# stats (200 steps), FETCH/WRITE PMC passes (30 steps), SQ counters (20 steps), chain block-count scaling, training-step bench
R=${R:-r06}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repo root)}"
python -m pytest tests -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_parity_gpu.py -m gpu -s -q -k "c2 or bf16 or drift" > gpurun_out/${R}_parity_at_benchmarked_config.log 2>&1; tail -2 gpurun_out/${R}_parity_at_benchmarked_config.log
if [ -z "$SKIP_TRAIN" ]; then
python -m pytest tests/test_train_step_gpu.py -m gpu -s -q 2>&1 | grep -E "\[f32\]|\[bf16|passed|failed" > gpurun_out/${R}_train_step_parity.log; tail -3 gpurun_out/${R}_train_step_parity.log
fi
python bench.py 2>gpurun_out/bench_default_err.log > gpurun_out/${R}_bench_full_1000steps.json; tail -c 2500 gpurun_out/${R}_bench_full_1000steps.json
python tools/chain_full_bench.py --forms 8 --reps 3 2>/dev/null | grep "waves:" > gpurun_out/${R}_chain_block_scaling.txt; cat gpurun_out/${R}_chain_block_scaling.txt
if [ -z "$SKIP_TRAIN" ]; then for b in 4 32; do python tools/train_bench.py --batch $b --iters 8 --kernels 2>/dev/null | tail -1; done > gpurun_out/${R}_train_step.jsonl; cut -c1-400 gpurun_out/${R}_train_step.jsonl; fi
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step --no-other-configs"
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 $ARGS --ddpm-steps 200 > gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_bench_200steps.csv; head -16 "$f" | cut -c1-160
rm -rf gpurun_out/prof_trace
# config 4 (5 dancers x 300 frames, 4 clips: the 1 500-key two-chunk self-attention inside the launch) under the same trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -- python3 $ARGS --batch 4 --dancers 5 --frames 300 --ddpm-steps 60 > gpurun_out/prof_c4.log 2>&1
f=$(find gpurun_out/prof_c4 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_config4_5x300_b4_60steps.csv; head -8 "$f" | cut -c1-160
rm -rf gpurun_out/prof_c4
# (roofline: bench.py runs the kernel-trace / FETCH_SIZE / WRITE_SIZE child passes itself; the default run above carries them)
SETS="SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_BUSY_CYCLES" bash tools/gpu_pmc2.sh > /dev/null 2>&1
cp gpurun_out/pmc2_summary.txt gpurun_out/${R}_pmc_SQ_counters_20steps.txt; head -8 gpurun_out/${R}_pmc_SQ_counters_20steps.txt | cut -c1-250
if [ -z "$SKIP_TRAIN" ]; then
# the training step under rocprofv3 (kernel stats of 6 steps at batch 32)
rm -rf gpurun_out/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_bench.py --batch 32 --iters 4 > gpurun_out/prof_train.log 2>&1
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${R}_kernel_stats_train_step_b32.csv; head -12 "$f" | cut -c1-160
rm -rf gpurun_out/prof_train
# per-call-site / per-shape tables of the training step
python tools/train_shapes.py --batch 32 --top 60 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_train_shapes_b32.log
python tools/gemm_shapes.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_gemm_shapes.log
python tools/attn_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_attn_bench.log
python tools/tn_probe.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_tn_probe.log
python tools/gemm_rows_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${R}_gemm_rows_bench.txt
# the training step with the row-block GEMM on / off, same box, interleaved
for rep in 1 2 3; do for rows in 1 0; do
  echo "TCDIFF_TRAIN_ROWS=$rows: $(TCDIFF_TRAIN_ROWS=$rows python tools/train_bench.py --batch 32 --iters 10 2>/dev/null | tail -1 | cut -c88-150)"
done; done > gpurun_out/${R}_train_rows_ab.txt; cat gpurun_out/${R}_train_rows_ab.txt
fi
# round 4 additions: in-kernel stamps + shader clock of the fused layer, small-batch modes, pure-load ceiling of the weight stream
[ -f tools/probe/libtc_STAMP.so ] || bash tools/ab_build.sh STAMP "-DCH_STAMP" > /dev/null 2>&1
TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_chain_stamps.txt; grep -E "fused layer|last wave|shader clock" gpurun_out/${R}_chain_stamps.txt
# round 5: the launch with the self-attention inside (blocks cut per sequence: 8 / 256 blocks), its forms against each other, and the sampler with / without it
SA=1 TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_chain_stamps_self_attention.txt; grep -E "fused layer|last wave|shader clock|self-attention" gpurun_out/${R}_chain_stamps_self_attention.txt
timeout 300 python tools/chain_sa_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_chain_self_attention_forms.txt; cat gpurun_out/${R}_chain_self_attention_forms.txt
for rep in 1 2; do for f in 0 1; do
  echo "TCDIFF_FUSE_SA=$f: $(TCDIFF_FUSE_SA=$f timeout 900 python bench.py --steps 30 --warmup 4 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs 2>/dev/null | tail -1 | cut -c80-190)"
done; done > gpurun_out/${R}_sampler_self_attention_ab.txt; cat gpurun_out/${R}_sampler_self_attention_ab.txt
timeout 1200 python tools/small_batch.py 2>&1 | tail -3 > gpurun_out/${R}_small_batch.txt; cat gpurun_out/${R}_small_batch.txt
python -m pytest tests/test_parity_gpu.py -m gpu -s -q -k "attribution" 2>&1 | grep -E "guided evaluation|  bf16|  f32 parity|passed|failed" >> gpurun_out/${R}_parity_at_benchmarked_config.log
timeout 120 python tools/power_watch.py 10 > gpurun_out/${R}_power_watch.txt 2>&1; head -2 gpurun_out/${R}_power_watch.txt
