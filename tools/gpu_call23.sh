#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 600 -k "attention_d64 and not fp8" 2>&1 | tail -5 > $O/pytest_attn.log
tail -3 $O/pytest_attn.log
{
  python tools/kbench.py attn1 32 5 16384 pre
  python tools/kbench.py attn1 32 10 4096 pre
  python tools/kbench.py attn1 32 20 1024 pre
  python tools/kbench.py attn1 32 5 16384
  python tools/kbench.py attn1 32 5 16384 zero
} 2>&1 | grep -v amdgpu.ids > $O/attn_prefetch.log
cat $O/attn_prefetch.log
