#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python -m pytest tests/test_depth_gpu.py -m gpu -q -x -s --timeout 300 -k "v02_conditioner or embedder" 2>&1 | grep -v "^$" | tail -30 | tee $O/pytest_depth_v02.log
