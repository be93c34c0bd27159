#!/bin/bash
# fused self-attention (TCDIFF_FUSE_SA=1): full-size parity tests under the flag, then the sampler A/B on the same box, interleaved
mkdir -p gpurun_out
{
TCDIFF_FUSE_SA=1 timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -s -k "full_batch_16 or full_1000_step or partition_determinism or drift" 2>&1 | grep -v "^$" | tail -25
for rep in 1 2; do
  for f in 0 1; do
    echo "== TCDIFF_FUSE_SA=$f rep $rep"
    TCDIFF_FUSE_SA=$f timeout 600 python bench.py --steps 200 --warmup 20 --no-pmc 2>gpurun_out/fuse_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" || tail -5 gpurun_out/fuse_err.log
  done
done
} > gpurun_out/fuse_sa.log 2>&1
tail -60 gpurun_out/fuse_sa.log
