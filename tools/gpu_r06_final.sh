#!/bin/bash
# round 6, final pass on the committed tree: GPU suite, smoke, default bench, kernel stats of the same command, stamps, small jobs
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r06_gpu_suite_final.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/r06_smoke.log
python bench.py 2>gpurun_out/bench_default_err.log > gpurun_out/r06_bench_full_1000steps.json; tail -c 1200 gpurun_out/r06_bench_full_1000steps.json
timeout 900 python bench.py --steps 20 --warmup 2 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs 2>/dev/null | tail -1 | cut -c1-200 | tee gpurun_out/r06_bench_20jobs.txt
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-parity-mode --no-train-step --no-other-configs"
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 $ARGS --ddpm-steps 200 > gpurun_out/prof_trace.log 2>&1
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_kernel_stats_bench_200steps.csv; head -10 "$f" | cut -c1-120
rm -rf gpurun_out/prof_trace
SETS="SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS" bash tools/gpu_pmc2.sh > /dev/null 2>&1
cp gpurun_out/pmc2_summary.txt gpurun_out/r06_pmc_SQ_counters_20steps.txt; head -4 gpurun_out/r06_pmc_SQ_counters_20steps.txt | cut -c1-250
SA=1 TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_stamps_self_attention.txt; grep -E "fused layer|last wave|shader clock|self-attention" gpurun_out/r06_chain_stamps_self_attention.txt
timeout 300 python tools/chain_sa_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_self_attention_forms.txt; cat gpurun_out/r06_chain_self_attention_forms.txt
timeout 1200 python tools/small_batch.py 2>&1 | tail -3 > gpurun_out/r06_small_batch.txt; cat gpurun_out/r06_small_batch.txt
for b in 4 32; do python tools/train_bench.py --batch $b --iters 8 --kernels 2>/dev/null | tail -1; done > gpurun_out/r06_train_step.jsonl; cut -c1-200 gpurun_out/r06_train_step.jsonl
TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so timeout 300 python tools/split_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_split_stamps.txt; grep -E "part|partial stored|V  " gpurun_out/r06_split_stamps.txt | head -12
rm -rf gpurun_out/prof_small
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small -- python3 tools/small_job.py > gpurun_out/prof_small.log 2>&1
f=$(find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_kernel_stats_small_job_ddim50_1clip.csv; head -12 "$f" | cut -c1-140
rm -rf gpurun_out/prof_small
