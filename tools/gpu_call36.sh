#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests/test_depth_gpu.py tests/test_clip_gpu.py -m gpu -q -x -s --timeout 600 > $O/pytest_depth_clip_full.log 2>&1
grep -v "^$" $O/pytest_depth_clip_full.log | grep -i "passed\|failed\|error\|depth\|layer_\|DepthEmbedder\|assert" | tail -30
