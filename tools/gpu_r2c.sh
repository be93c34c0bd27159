#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r2c_all.log 2>&1; echo "all tests rc=$?"
grep -E "^F?\.?chain vs|bf16|passed|failed|Error" gpurun_out/r2c_all.log | grep -v "^ " | tail -40
