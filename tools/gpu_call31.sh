#!/bin/bash
# ffn2 timing ablations (HI3D_FFN2_ABL bits: 1 no weight DMA in the loop, 2 no GELU math, 4 no rolling W1 reads,
# 8 no rolling W2 reads, 16 no second GEMM, 32 no loop barriers)
O=gpurun_out/r02d; mkdir -p $O
for a in 0 1 2 4 8 12 16 32 33 47 63; do
  echo -n "abl=$a "; HI3D_FFN2_ABL=$a timeout 100 python tools/kbench.py ffn 2>&1 | grep "fused   M= 524288"
done | tee $O/ffn2_ablation.log
