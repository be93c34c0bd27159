#!/bin/bash
# round 5, visit d: issue-priority variants of the 8-wave chain kernel (fused-layer launch, same box, 2 interleaved rounds)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in BASE PRIO1 PRIO2 PRIO3; do
  echo "== $v"; TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_full_bench.py --forms 8 --blocks 1,225 --reps 3 2>&1 | grep "waves:"
done; done | tee gpurun_out/r05_prio_variants.txt
