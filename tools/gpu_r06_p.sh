#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python tools/fill_census.py 2>&1 | grep -v amdgpu.ids | tail -95 | tee gpurun_out/r06_train_fill_census.txt
