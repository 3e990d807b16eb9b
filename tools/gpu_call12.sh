#!/bin/bash
# round 2, call 12: ping-pong attention kernel -- parity (all attention tests in both modes, race screen), timing vs the 256-row kernel
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x --timeout 600 -k "attention_d64 and not fp8" 2>&1 | tail -12 > $O/pytest_attn_pp.log
tail -6 $O/pytest_attn_pp.log
{
for pp in 0 1; do
  echo "== HI3D_ATTN_PP=$pp"
  HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 5 16384 pre
  HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 10 4096 pre
  HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 20 1024 pre
  HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 5 4096 pre
  HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 5 16384
done
} 2>&1 | grep -v amdgpu.ids > $O/attn_pp.log
cat $O/attn_pp.log
