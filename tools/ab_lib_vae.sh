#!/bin/bash
# A/B two builds of the library on ONE box for the VAE decoder: gpurun_tmp/lib_a.so and gpurun_tmp/lib_b.so alternated ${AB_REPS:-3}
# times under tools/vae_bench.py.   usage: bash tools/ab_lib_vae.sh <out.log> <label_a> <label_b>
OUT=$1; LA=${2:-a}; LB=${3:-b}
L=hi3d-official_amd/hi3d_hip/libhi3d_hip.so
cp $L /tmp/keep.so
: > $OUT
for rep in $(seq 1 ${AB_REPS:-3}); do
  for v in a b; do
    cp gpurun_tmp/lib_$v.so $L
    [ $v = a ] && label=$LA || label=$LB
    echo "== $label run $rep" >> $OUT
    python tools/vae_bench.py 2> /dev/null >> $OUT
  done
done
cp /tmp/keep.so $L
cat $OUT
