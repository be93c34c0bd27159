#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
F="--steps 12 --warmup 2 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3 4; do for f in 0 1; do
  echo "TCDIFF_FORK_PROLOGUE=$f: $(TCDIFF_FORK_PROLOGUE=$f timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c80-120)"
done; done | tee gpurun_out/r06_prologue_fork_ab.txt
