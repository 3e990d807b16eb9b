#!/bin/bash
O=gpurun_out/r02b; mkdir -p $O
{
for fl in 0 1 2 3; do
  echo "== PP=1 HI3D_ATTN_DEPHASE(flags)=$fl"
  HI3D_ATTN_PP=1 HI3D_ATTN_DEPHASE=$fl,0 python tools/kbench.py attn1 32 5 16384 pre
  HI3D_ATTN_PP=1 HI3D_ATTN_DEPHASE=$fl,0 python tools/kbench.py attn1 32 10 4096 pre
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_pp_prio.log
cat $O/attn_pp_prio.log
