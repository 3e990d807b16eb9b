#!/bin/bash
# round 2, call 10: attention timing ablations (what does a key tile spend its time on?)
O=gpurun_out/r02b
mkdir -p $O
{
for abl in "" 0 1 2 3 4 5; do
  echo "== HI3D_ATTN_ABL=$abl"
  if [ -z "$abl" ]; then python tools/kbench.py attn1 32 5 16384 pre; else HI3D_ATTN_ABL=$abl python tools/kbench.py attn1 32 5 16384 pre; fi
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_ablation.log
cat $O/attn_ablation.log
