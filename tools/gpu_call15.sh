#!/bin/bash
O=gpurun_out/r02b; mkdir -p $O
{
for abl in 0 1 5; do
  echo "== PP=1 ABL=$abl"
  HI3D_ATTN_PP=1 HI3D_ATTN_ABL=$abl python tools/kbench.py attn1 32 5 16384 pre
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_pp_abl.log
cat $O/attn_pp_abl.log
