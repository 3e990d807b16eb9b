#!/bin/bash
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${tag}_$i -- "$@" > /tmp/pmc_${tag}_$i.log 2>&1)
  python $R/tools/pmc_sum.py /tmp/pmc_${tag}_$i "$PMC_FILTER"
done
