#!/bin/bash
# usage: pmc.sh <tag> <cmd...>   -- runs SQ counter passes on the command, prints sums
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_LDS SQ_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${tag}_$i -- "$@" > /tmp/pmc_${tag}_$i.log 2>&1)
done
python $R/tools/pmc_sum.py "/tmp/pmc_${tag}_*" "$PMC_FILTER" 2>/dev/null || for j in 1 2 3 4 5 6; do python $R/tools/pmc_sum.py /tmp/pmc_${tag}_$j "$PMC_FILTER"; done
