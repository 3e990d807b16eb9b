#!/bin/bash
# A/B of environment switches on ONE box (boxes of the pool differ by +-1.5 %): alternates the settings, twice.
#   usage: bash tools/ab_env.sh <out.log> "<label_a>:<ENV=..;ENV=..>" "<label_b>:<...>" ...     (an empty setting = the default tree)
# Each run is the driver-form headline (bench.py --no-legs --no-traffic --no-cpu-baseline): wall ms/step of the graph-replayed step
# and the per-family kernel times of the eager profile pass.
OUT=$1; shift
: > $OUT
for rep in $(seq 1 ${AB_REPS:-2}); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    ( IFS=';'; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
      HI3D_BENCH_PARITY=0 python bench.py --steps 10 --warmup 3 --no-legs --no-traffic --no-cpu-baseline 2> /tmp/ab_err.log > /tmp/ab_out.json
      python - "$label" "$rep" <<'PY' >> $OUT
import json, sys
d = json.load(open("/tmp/ab_out.json"))
k = d.get("kernels_ms_per_step", {})
print(f"{sys.argv[1]:28s} run {sys.argv[2]}: {d['ms_per_step']:8.2f} ms/step   " + "  ".join(f"{a}={b:.2f}" for a, b in k.items()))
PY
    )
  done
done
cat $OUT
