#!/bin/bash
# round 6, visit A: the software-pipelined in-kernel attention (CH_ATT_PIPE): tests, the fused launch per library variant, stamps, sampler A/B
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repo root)}"

V="BASE SAONLY NOSGB ROWLAT"
for rep in 1 2; do
  echo -n "default : "; timeout 200 python tools/chain_sa_bench.py --reps 3 --only "self-attention" 2>&1 | grep "rows,"
  for v in $V; do
    echo -n "$v : "; TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 200 python tools/chain_sa_bench.py --reps 3 --only "self-attention" 2>&1 | grep "rows,"
  done
done
for v in STAMP0 STAMP; do
  SA=1 TCDIFF_LIB_PATH=tools/probe/libtc_$v.so timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_chain_stamps_$v.txt
  grep -E "fused layer|last wave|shader clock|self-attention|cross-attention" gpurun_out/r06_chain_stamps_$v.txt
done
F="--steps 3 --warmup 1 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2; do
  echo -n "sampler default: "; timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c1-140
  echo -n "sampler BASE   : "; TCDIFF_LIB_PATH=tools/probe/libtc_BASE.so timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c1-140
done
