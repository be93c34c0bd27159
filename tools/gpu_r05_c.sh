#!/bin/bash
# round 5, visit c: in-kernel stamps of the fused layer, 8-wave vs 4-wave form
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nw in 8 4; do
  NW=$nw TCDIFF_LIB_PATH=tools/probe/libtc_STAMP.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_chain_stamps_nw$nw.txt
done
paste -d'|' <(cut -c1-62 gpurun_out/r05_chain_stamps_nw8.txt) <(cut -c52-62 gpurun_out/r05_chain_stamps_nw4.txt) | head -60
