#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
echo "--- default"; timeout 200 python tools/diag_attn_ragged.py 2>&1 | grep -v amdgpu.ids | tee $O/diag_attn_ragged_default.log
echo "--- HI3D_ATTN_FORCE_EXACT=1"; HI3D_ATTN_FORCE_EXACT=1 timeout 200 python tools/diag_attn_ragged.py 2>&1 | grep -v amdgpu.ids | tee $O/diag_attn_ragged_force_exact.log
