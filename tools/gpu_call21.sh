#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_vae_gpu.py -q -x --timeout 900 2>&1 | tail -4 > $O/pytest_unet_vae.log
tail -3 $O/pytest_unet_vae.log
python bench.py --config s2 --steps 10 --warmup 3 --shapes --no-cpu-baseline > $O/s2_bench2.json 2> $O/s2_bench2.log
cut -c1-330 $O/s2_bench2.json; head -13 $O/s2_bench2.log
