#!/bin/bash
# round 2, call 9: are the two co-resident attention blocks of a CU in phase?  (delay every other block at start)
set -x
O=gpurun_out/r02b
mkdir -p $O
{
for cfg in "" "1,1" "1,2" "1,4" "2,1" "2,2" "2,4" "3,2"; do
  echo "== HI3D_ATTN_DEPHASE=$cfg"
  HI3D_ATTN_DEPHASE=$cfg python tools/kbench.py attn1 32 5 16384
  HI3D_ATTN_DEPHASE=$cfg python tools/kbench.py attn1 32 10 4096
done
} > $O/attn_dephase.log 2>&1
cat $O/attn_dephase.log
