#!/bin/bash
# round 2, call 14: effective clock and SQ busy counters of the two attention kernels (is MFMA || VALU overlap paid back as clock?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for pp in 0 1; do
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rm -rf /tmp/pmc_a
    HI3D_ATTN_PP=$pp rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_a -o run --output-format csv -- python $R/tools/kbench.py attn1 32 5 16384 pre > $O/pmc_attn_pp${pp}_$i.log 2>&1
    echo "== PP=$pp set $i" >> $O/attn_pmc.csv
    python $R/tools/pmc_sum.py /tmp/pmc_a attn_d64 >> $O/attn_pmc.csv 2>&1
    python - <<PY >> $O/attn_pmc.csv
import csv, glob
for f in glob.glob('/tmp/pmc_a/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'attn_d64' in r.get('Kernel_Name','')]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    if d: print('kernel_trace durations us:', ' '.join(f'{x:.0f}' for x in d))
PY
  done
done
cat $O/attn_pmc.csv
