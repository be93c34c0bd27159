#!/bin/bash
# round 6, visit C: what the in-kernel attention loop is made of -- stamps of the pipelined loop with one ingredient removed at a time
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in STAMP0 STAMP ST_MMA ST_EXP ST_CVT ST_MAX ST_LD ST_VALU ST_ALL; do
  echo "== $v"
  SA=1 TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -E "fused layer|shader clock|self-attention|cross-attention"
done
