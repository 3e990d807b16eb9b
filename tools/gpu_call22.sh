#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
{
for abl in 0 1 5 9 13 2; do
 echo "== variant 7 ABL=$abl"
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one geglu 131072 5120 640
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one plain 524288 960 320
done
} 2>&1 | grep -v amdgpu.ids > $O/gemm_epilogue_ablation3.log
cat $O/gemm_epilogue_ablation3.log
