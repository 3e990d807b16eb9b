#!/bin/bash
# round 2, call 8: wide-tile residual prefetch; MFMA/VALU co-issue probe
set -x
O=gpurun_out/r02b
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x --timeout 600 -k "gemm or conv" 2>&1 | tail -5 > $O/pytest_gemm2.log
tail -3 $O/pytest_gemm2.log
timeout 600 python tools/kbench.py sweep h 0 7 > $O/sweep3.log 2>&1
cat $O/sweep3.log
timeout 120 ./gpurun_tmp/coissue > $O/coissue.log 2>&1
cat $O/coissue.log
