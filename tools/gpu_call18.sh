#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
{
for v in 5 7; do
 echo "== variant $v"
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one geglu 131072 5120 640
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one plain 131072 5120 640
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one res 131072 5120 640
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one geglu 32768 10240 1280
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one plain 32768 10240 1280
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one geglu 131072 5120 2560
 HI3D_GEMM_VARIANT=$v python tools/kbench.py one plain 131072 5120 2560
done
} 2>&1 | grep -v amdgpu.ids > $O/geglu_epilogue.log
cat $O/geglu_epilogue.log
