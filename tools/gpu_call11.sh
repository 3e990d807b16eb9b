#!/bin/bash
O=gpurun_out/r02b
mkdir -p $O
python tools/occ_check.py 2>&1 | grep -v amdgpu.ids > $O/occupancy.log
for bh in "1 4" "1 8" "1 16"; do python tools/kbench.py attn1 $bh 16384 pre 2>&1 | grep -v amdgpu.ids >> $O/occupancy.log; done
cat $O/occupancy.log
