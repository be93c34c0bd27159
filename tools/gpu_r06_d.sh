#!/bin/bash
# round 6, visit D: full GPU suite on the tree (hazard-checked lane maxima, two-part step prologue), then the prologue fork A/B
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r06_gpu_suite_d.log
F="--steps 4 --warmup 1 --no-pmc --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in 1 2 3; do for f in 0 1; do
  echo "TCDIFF_FORK_PROLOGUE=$f: $(TCDIFF_FORK_PROLOGUE=$f timeout 600 python bench.py $F 2>gpurun_out/ab_err.log | tail -1 | cut -c1-120)"
done; done | tee gpurun_out/r06_prologue_fork_ab.txt
