#!/bin/bash
# A/B two builds of the library on ONE box: gpurun_tmp/lib_a.so and gpurun_tmp/lib_b.so, alternated ${AB_REPS:-3} times; driver-form
# headline without legs.   usage: bash tools/ab_lib.sh <out.log> <label_a> <label_b>
OUT=$1; LA=${2:-a}; LB=${3:-b}
L=hi3d-official_amd/hi3d_hip/libhi3d_hip.so
cp $L /tmp/keep.so
: > $OUT
for rep in $(seq 1 ${AB_REPS:-3}); do
  for v in a b; do
    cp gpurun_tmp/lib_$v.so $L
    [ $v = a ] && label=$LA || label=$LB
    HI3D_BENCH_PARITY=0 python bench.py --steps 10 --warmup 3 --no-legs --no-traffic --no-cpu-baseline 2> /dev/null > /tmp/ab_out.json
    python - "$label" "$rep" <<'PY' >> $OUT
import json, sys
d = json.load(open("/tmp/ab_out.json"))
k = d.get("kernels_ms_per_step", {})
print(f"{sys.argv[1]:28s} run {sys.argv[2]}: {d['ms_per_step']:8.2f} ms/step   " + "  ".join(f"{a}={b:.2f}" for a, b in k.items()))
PY
  done
done
cp /tmp/keep.so $L
cat $OUT
