#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 600 -k "layernorm or layer_norm or norm" 2>&1 | tail -3
echo "== packed"; python tools/kbench.py norm 2>&1 | grep -v amdgpu.ids | tail -4
echo "== one wave per row"; HI3D_LN_PACKED=0 python tools/kbench.py norm 2>&1 | grep -v amdgpu.ids | tail -4
