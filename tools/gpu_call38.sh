#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
timeout 300 python tools/diag_attn_ragged.py 2>&1 | grep -v amdgpu.ids | tee $O/diag_attn_ragged.log | tail -40
