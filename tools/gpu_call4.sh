#!/bin/bash
set -x
mkdir -p gpurun_out/r02d
./gpurun_tmp/mx_scale_layout > gpurun_out/r02d/mx_probe.log 2>&1
python -m pytest tests/test_kernels_gpu.py -q -x --timeout 600 -s -k "fp8qk" 2>&1 | tail -15 > gpurun_out/r02d/pytest_fp8.log
python -m pytest tests/test_unet_gpu.py -q -x --timeout 600 -s -k "fp8" 2>&1 | tail -8 >> gpurun_out/r02d/pytest_fp8.log
for mb in 0 96 176; do echo "== HI3D_GN_SPLIT_MB=$mb"; HI3D_GN_SPLIT_MB=$mb python tools/kbench.py norm; done > gpurun_out/r02d/gn_split.log 2>&1
cat gpurun_out/r02d/mx_probe.log gpurun_out/r02d/pytest_fp8.log gpurun_out/r02d/gn_split.log
