#!/bin/bash
# round 6, visit V: host side of a job -- cached step tables, host-to-device copies before the job's device work
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/small_job_setup.py 2>&1 | grep -E "^job|profiled|device kernels|setup kernels" | tee gpurun_out/r06_small_job_setup.txt
timeout 900 python tools/small_batch.py 2 2>&1 | tail -1 | tee gpurun_out/r06_small_batch_split.txt
