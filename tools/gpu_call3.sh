#!/bin/bash
set -x
mkdir -p gpurun_out/r02c
python -m pytest tests/test_kernels_gpu.py -q -x --timeout 600 -s -k "fp8qk" 2>&1 | tail -15 > gpurun_out/r02c/pytest_fp8.log
python -m pytest tests/test_unet_gpu.py -q -x --timeout 600 -s -k "fp8" 2>&1 | tail -8 >> gpurun_out/r02c/pytest_fp8.log
python -m pytest tests/test_parallel_gpu.py -q -x --timeout 600 -s 2>&1 | tail -30 > gpurun_out/r02c/pytest_parallel.log
python bench.py --steps 6 --warmup 3 --attn fp8qk --no-cpu-baseline > gpurun_out/r02c/s2_fp8qk.json 2> gpurun_out/r02c/s2_fp8qk.log
cat gpurun_out/r02c/pytest_fp8.log; tail -14 gpurun_out/r02c/pytest_parallel.log
grep "\[bench\]" gpurun_out/r02c/s2_fp8qk.log | head -14
