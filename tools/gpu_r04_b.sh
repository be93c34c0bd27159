#!/bin/bash
# ablation / MFMA-shape timing experiments on the fused layer chain (stamp builds): where do the cycles and the clock go?
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for v in STAMP S_MFMA16 S_NOGELU S_NOXATT S_NOST STAMP; do
  echo "==== $v"
  TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stamps_$v.txt
  grep -E "fused layer|last wave|shader clock" gpurun_out/r04_stamps_$v.txt
done
