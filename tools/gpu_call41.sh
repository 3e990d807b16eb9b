#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 100 python -m pytest tests/test_unet_gpu.py -m gpu -q -x --timeout 90 --durations=6 > $O/pytest_unet_final.log 2>&1
tail -14 $O/pytest_unet_final.log
