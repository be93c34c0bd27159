#!/bin/bash
# in-launch self-attention (TCDIFF_FUSE_SA=1): kernel tests, full-size parity tests under the flag, then the sampler A/B on the same
# box, interleaved (REPS pairs of STEPS-step runs)
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_chain_selfatt_gpu.py tests/test_chain_gpu.py -x -q 2>&1 | tail -2
TCDIFF_FUSE_SA=1 timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -s -k "full_batch_16 or full_1000_step or partition_determinism or drift or c2_bf16" 2>&1 | grep -v "^$" | tail -${TAILN:-12}
for rep in $(seq 1 ${REPS:-3}); do
  for f in 0 1; do
    echo "== TCDIFF_FUSE_SA=$f rep $rep"
    TCDIFF_FUSE_SA=$f timeout 900 python bench.py --steps ${STEPS:-120} --warmup 10 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs 2>gpurun_out/fuse_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || tail -5 gpurun_out/fuse_err.log
  done
done
} > gpurun_out/fuse_sa.log 2>&1
tail -60 gpurun_out/fuse_sa.log
