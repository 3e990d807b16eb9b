#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 600 -k "gemm or conv or ffn or dense" 2>&1 | tail -5 > $O/pytest_gemm3.log
tail -3 $O/pytest_gemm3.log
{
for abl in 0 1; do
 echo "== variant 7 ABL=$abl (per-wave epilogue)"
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one geglu 131072 5120 640
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one plain 131072 5120 640
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one plain 524288 960 320
 HI3D_GEMM_ABL=$abl HI3D_GEMM_VARIANT=7 python tools/kbench.py one res 131072 640 2560
done
} 2>&1 | grep -v amdgpu.ids > $O/gemm_epilogue_ablation2.log
cat $O/gemm_epilogue_ablation2.log
timeout 600 python tools/kbench.py sweep h 0 > $O/sweep4.log 2>&1
cat $O/sweep4.log
