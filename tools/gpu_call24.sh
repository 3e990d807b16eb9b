#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
{
echo "== fused FFN random"; python tools/kbench.py ffn
echo "== fused FFN zero operands"; python tools/kbench.py ffn zero
for d in rand zero; do
  echo "== data=$d"
  python tools/kbench.py one geglu 131072 5120 640 $d
  python tools/kbench.py one plain 524288 960 320 $d
  python tools/kbench.py one res 131072 640 2560 $d
done
} 2>&1 | grep -v amdgpu.ids > $O/power_probe2.log
cat $O/power_probe2.log
