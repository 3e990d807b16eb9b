#!/bin/bash
# round 2, call 16: is the attention / GEMM rate set by the power budget?  (same kernels on random, constant and zero operands)
O=gpurun_out/r02b; mkdir -p $O
{
for d in pre const zero; do
  for pp in 0 1; do
    echo "== attention PP=$pp data=$d"; HI3D_ATTN_PP=$pp python tools/kbench.py attn1 32 5 16384 $d
  done
done
for d in rand zero; do
  echo "== conv-like dense GEMM M=32768 N=1280 K=5120 res, data=$d"; python tools/kbench.py one res 32768 1280 5120 $d
  echo "== geglu M=131072 N=5120 K=640, data=$d"; python tools/kbench.py one geglu 131072 5120 640 $d
done
} 2>&1 | grep -v amdgpu.ids > $O/power_probe.log
cat $O/power_probe.log
