#!/bin/bash
# round-4 first visit: L2 aliasing probe, stream probe, stamps of the fused layer, chain block scaling (baseline)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 tools/probe/l2_alias_probe > gpurun_out/r04_l2_alias_probe.txt 2>&1; echo "alias rc=$?"
timeout 300 tools/probe/stream_probe > gpurun_out/r04_stream_probe.txt 2>&1; echo "stream rc=$?"
timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_chain_stamps_base.txt; echo "stamps rc=$?"
timeout 300 python tools/chain_bench.py 2>/dev/null | grep "chain" > gpurun_out/r04_chain_block_scaling_base.txt
cat gpurun_out/r04_l2_alias_probe.txt
