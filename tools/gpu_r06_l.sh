#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so timeout 300 python tools/split_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_split_stamps.txt
LQ=60 NSEQ=2 LK=62 TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so timeout 300 python tools/split_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_split_stamps_c1.txt
