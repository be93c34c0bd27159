#!/bin/bash
# in-launch self-attention at every job size: kernel tests, ALL parity tests under the flag, small-batch latency with and without
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_chain_selfatt_gpu.py -x -q 2>&1 | tail -2
TCDIFF_FUSE_SA=1 timeout 2400 python -m pytest tests/test_parity_gpu.py -q 2>&1 | tail -12
for f in 0 1; do echo "== small batch, TCDIFF_FUSE_SA=$f"; TCDIFF_FUSE_SA=$f timeout 900 python tools/small_batch.py 2 2>&1 | tail -1; done
} > gpurun_out/fuse_sa2.log 2>&1
tail -40 gpurun_out/fuse_sa2.log
