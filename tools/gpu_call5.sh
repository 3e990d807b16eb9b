#!/bin/bash
set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x --timeout 900 -k "not clip_parallel" 2>&1 | tail -15 > gpurun_out/r02/pytest_gpu.log
tail -5 gpurun_out/r02/pytest_gpu.log
for v in 0 2; do echo "== HI3D_GEMM_VARIANT=$v (conv)"; HI3D_GEMM_VARIANT=$v python tools/kbench.py conv; done > gpurun_out/r02/conv_variants.log 2>&1
cat gpurun_out/r02/conv_variants.log
bash tools/profile_round.sh r02
