#!/bin/bash
# round 5, visit a: the RCCL world-1 tests, the DDIM no-clip golden test, the padded co-issue probe
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_rccl_world1_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r05_rccl_world1.log; tail -5 gpurun_out/r05_rccl_world1.log
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "ddim_sample_honours or constructor_options" 2>&1 | grep -E "ddim_sample clip|loop_|passed|failed|Error" | tail -12
timeout 300 ./tools/probe/coissue_probe > gpurun_out/r05_coissue_probe.txt 2>&1; grep "padded probe" gpurun_out/r05_coissue_probe.txt
