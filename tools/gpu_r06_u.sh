#!/bin/bash
# round 6, visit U: the four-wave form of the chain launches in the SUSTAINED sampler (the round's energy-bound reading: half the LDS
# fragment reads per block), interleaved with the default eight-wave form on one box
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
ARGS="bench.py --steps 10 --warmup 2 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for i in 1 2 3; do
  for nw in 8 4; do
    v=$(TCDIFF_CHAIN_NW=$nw timeout 900 python $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "TCDIFF_CHAIN_NW=$nw: $v clips/s"
  done
done | tee gpurun_out/r06_chain_nw_sustained_ab.txt
