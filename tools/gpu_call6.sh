#!/bin/bash
# round 2, call 6: ping-pong GEMM variants (6: 256x160/128 3-stage, 7: 256x320 2-stage) -- parity, race screen, per-shape sweep
set -x
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x --timeout 600 -k "gemm or conv" 2>&1 | tail -15 > gpurun_out/r02b/pytest_gemm.log
tail -8 gpurun_out/r02b/pytest_gemm.log
timeout 900 python tools/kbench.py sweep > gpurun_out/r02b/sweep.log 2>&1
cat gpurun_out/r02b/sweep.log
