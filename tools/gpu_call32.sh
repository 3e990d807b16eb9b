#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -m gpu -q -x -k ffn --timeout 120 2>&1 | tail -5
echo "--- ffn2"; timeout 200 python tools/kbench.py ffn 2>&1 | grep fused | tee $O/kbench_ffn2_v2.log
