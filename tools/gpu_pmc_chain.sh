#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_chain$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_chain$i -- python3 tools/chain_only.py > gpurun_out/pmc_chain$i.log 2>&1; echo "rc=$?"
  python3 tools/pmc_summary.py gpurun_out/pmc_chain$i | grep -E "chain_tail|gemm_rowln|gemm_tile" | sed 's/  */ /g'
done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
