#!/bin/bash
# Diagnostic builds of the library with one ingredient of the chain GEMM loops removed (results are wrong on purpose):
# which of weight loads / LDS fragment reads / MFMAs / the L2 prefetch sets the loop time?  Run: tools/chain_ablate.sh build
# here (cross-compiles), then `tools/chain_ablate.sh run` on the GPU box.
cd "$(dirname "$0")/.."
V="STAMP"
if [ "$1" = build ]; then
  for v in $V; do
    flags=""; for p in ${v//_/ }; do flags="$flags -DCH_ABLATE_$p"; done
    [ $v != PF ] && flags="$flags -DCH_ABLATE_PF"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itcdiff_amd/csrc $flags -shared -o tools/probe/libtc_$v.so \
      tcdiff_amd/csrc/chain.hip tcdiff_amd/csrc/ops.hip tcdiff_amd/csrc/gemm.hip tcdiff_amd/csrc/attention.hip 2>&1 | grep -E "error" &
  done; wait; ls -la tools/probe/*.so
else
  python tools/chain_stamps.py
fi
