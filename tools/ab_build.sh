#!/bin/bash
# A/B builds of the library: tools/ab_build.sh NAME "-DFLAG ..."  ->  tools/probe/libtc_NAME.so
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itcdiff_amd/csrc $2 -shared \
  -o tools/probe/libtc_$1.so tcdiff_amd/csrc/*.hip 2>&1 | grep -E "error"
ls -la tools/probe/libtc_$1.so
