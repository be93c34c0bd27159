#!/bin/bash
# A/B on one box: cross-attention with two passes over K / V (TP) against one pass (SP)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in TP_STAMP SP_STAMP TP_STAMP SP_STAMP; do
  echo "==== $v"
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stamps_$v.txt
  grep -E "last wave|shader clock|cross-attention" gpurun_out/r04_stamps_$v.txt
done
bash tools/ab_run.sh TP SP
