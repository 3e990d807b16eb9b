#!/bin/bash
# A/B two builds of the library on ONE box (boxes of the pool differ by several percent): put the two
# builds at gpurun_tmp/lib_a.so and gpurun_tmp/lib_b.so, run this through gpurun; it alternates the
# builds twice and prints the per-family ms/step of bench.py.
L=hi3d-official_amd/hi3d_hip/libhi3d_hip.so
cp $L /tmp/keep.so
for rep in 1 2; do
  for v in a b; do
    cp gpurun_tmp/lib_$v.so $L
    echo "== $v run $rep"
    python bench.py --config s2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "\[bench\] (gemm|attn_d64|group|kernels)" | awk '{print $2, $3}' | tr '\n' ' '; echo
  done
done
cp /tmp/keep.so $L
