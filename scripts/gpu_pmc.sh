#!/bin/bash
# PMC passes over tools/microbench.py: MFMA utilisation / stall breakdown of the hot kernels.
export PYTHONUNBUFFERED=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
# PMC_SETS=short: the three passes the summary (tools/pmc_summary.py) needs
if [ "$PMC_SETS" == "short" ]; then
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS")
else
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
        "GRBM_GUI_ACTIVE GRBM_COUNT")
fi
: > $R/gpurun_out/pmc/status.log
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/set$i -o p -- python $R/tools/microbench.py > $R/gpurun_out/pmc/set$i.log 2>&1
  echo "set$i ($set) rc=$?" >> $R/gpurun_out/pmc/status.log
done
cat $R/gpurun_out/pmc/status.log
