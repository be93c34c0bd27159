#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cfg in "150 3 4" "130 3 4" "450 2 4" "150 3 2"; do
  for v in BASE NOSGB DEF; do
    lib=tools/probe/libtc_$v.so; [ $v = DEF ] && lib=tcdiff_amd/libtcdiff_gfx950.so
    TCDIFF_LIB_PATH=$lib python tools/chain_sa_dump.py /tmp/d_$v.pt $cfg 2>&1 | grep saved
  done
  echo "== $cfg: BASE vs NOSGB"; python tools/chain_sa_cmp.py /tmp/d_BASE.pt /tmp/d_NOSGB.pt
  echo "== $cfg: BASE vs DEF"; python tools/chain_sa_cmp.py /tmp/d_BASE.pt /tmp/d_DEF.pt
done
timeout 1200 python -m pytest tests/test_chain_selfatt_gpu.py tests/test_chain_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -12
bash tools/gpu_r06_a.sh 2>&1 | tail -40
