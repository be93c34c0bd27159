#!/bin/bash
# round-2 first visit: GPU tests (new parity cases print their observed errors), default bench, 2-rank bench through the
# launcher on one GPU box is not possible (1 GPU): the launcher itself is covered on CPU.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -s -k "bf16 or full_batch_16 or drift or boundary or resident" > gpurun_out/r2a_newtests.log 2>&1; echo "new tests rc=$?"
grep -E "vs reference|bf16 vs f32|passed|failed|Error|error" gpurun_out/r2a_newtests.log | tail -40
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_alltests.log 2>&1; echo "all tests rc=$?"; tail -3 gpurun_out/r2a_alltests.log
timeout 900 python bench.py > gpurun_out/r2a_bench.log 2>&1; echo "bench rc=$?"; tail -c 3000 gpurun_out/r2a_bench.log
