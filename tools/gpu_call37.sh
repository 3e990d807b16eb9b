#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests/test_depth_gpu.py -m gpu -q -s --timeout 600 -k "at_size or v02_conditioner" > $O/pytest_depth_tail.log 2>&1
grep -v "^$" $O/pytest_depth_tail.log | grep -i "passed\|failed\|error\|depth\|assert" | tail -20
