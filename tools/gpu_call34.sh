#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python -m pytest tests/test_depth_gpu.py -m gpu -q -x -s --timeout 300 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest_depth.log
