#!/bin/bash
# Diagnostic build of the library with per-phase timestamps in the chain kernel (-DCH_STAMP; block 0 / wave 0 writes the
# 100 MHz wall counter at every phase boundary into h_out).  `tools/chain_ablate.sh build` here (cross-compiles), then
# `tools/chain_ablate.sh run` on the GPU box prints the time line (tools/chain_stamps.py).  Extra -D flags: $CH_FLAGS.
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itcdiff_amd/csrc -DCH_STAMP $CH_FLAGS -shared \
    -o tools/probe/libtc_STAMP.so tcdiff_amd/csrc/chain.hip tcdiff_amd/csrc/ops.hip tcdiff_amd/csrc/gemm.hip \
    tcdiff_amd/csrc/attention.hip tcdiff_amd/csrc/train.hip 2>&1 | grep -E "error"
  ls -la tools/probe/*.so
else
  python tools/chain_stamps.py
fi
