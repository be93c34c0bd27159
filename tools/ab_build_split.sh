#!/bin/bash
# A/B builds that differ in chain_split.hip only: tools/ab_build_split.sh NAME "-DFLAG ..."  ->  tools/probe/libtc_NAME.so
cd "$(dirname "$0")/.."
mkdir -p /tmp/abobj tools/probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itcdiff_amd/csrc $2 -c tcdiff_amd/csrc/chain_split.hip -o /tmp/abobj/split_$1.o 2>&1 | grep -E "error"
objs=$(ls tcdiff_amd/build/*.o | grep -v chain_split.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probe/libtc_$1.so /tmp/abobj/split_$1.o $objs
ls -la tools/probe/libtc_$1.so
