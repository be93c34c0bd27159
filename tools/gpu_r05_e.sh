#!/bin/bash
# round 5, visit e: self-attention softmax -- unpacked arithmetic and raised issue priority outside the PV stream (same box, 3 rounds)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in BASE ATT_U ATT_P ATT_UP ATT_UP1; do
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 120 python tools/attn_infer_bench.py 16 2>&1 | grep "workgroups"
done; done | tee gpurun_out/r05_attn_variants.txt
