#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TCDIFF_LIB_PATH=tools/probe/libtc_ATT_GLDS.so timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | tail -2
for rep in 1 2 3; do for v in BASE ATT_GLDS; do
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 120 python tools/attn_infer_bench.py 32 16 2>&1 | grep "workgroups"
done; done | tee gpurun_out/r05_attn_glds.txt
