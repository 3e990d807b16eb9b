#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_depth_gpu.py -m gpu -q -s --timeout 300 -k "ragged_repeatable or at_size or attention_d64" > $O/pytest_attn_ragged.log 2>&1
grep -v "^$" $O/pytest_attn_ragged.log | grep -i "passed\|failed\|error\|DepthEmbedder\|assert\|differ" | tail -12
