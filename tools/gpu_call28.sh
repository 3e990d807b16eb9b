#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 600 -k "ffn" 2>&1 | tail -3
python tools/kbench.py ffn 2>&1 | grep -v amdgpu.ids | tail -4
