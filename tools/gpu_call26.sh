#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 600 -k "temporal" 2>&1 | tail -3
python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids | tail -5
