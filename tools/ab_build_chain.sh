#!/bin/bash
# A/B builds that differ in chain.hip only: tools/ab_build_chain.sh NAME "-DFLAG ..."  ->  tools/probe/libtc_NAME.so
# (chain.hip is recompiled with the flags and linked with the other objects of the last `python -m tcdiff_amd.build`)
cd "$(dirname "$0")/.."
mkdir -p /tmp/abobj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itcdiff_amd/csrc $2 -c tcdiff_amd/csrc/chain.hip -o /tmp/abobj/chain_$1.o 2>&1 | grep -E "error"
objs=$(ls tcdiff_amd/build/*.o | grep -v chain.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libtc_$1.so /tmp/abobj/chain_$1.o $objs
ls -la tools/probe/libtc_$1.so
