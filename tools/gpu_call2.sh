#!/bin/bash
set -x
mkdir -p gpurun_out/r02b
python bench.py --steps 10 --warmup 3 --shapes --no-cpu-baseline > gpurun_out/r02b/s2_bench.json 2> gpurun_out/r02b/s2_bench.log
python bench.py --config s1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02b/s1_bench.json 2> gpurun_out/r02b/s1_bench.log
python bench.py --config vae --steps 2 --warmup 1 > gpurun_out/r02b/vae_bench.json 2> gpurun_out/r02b/vae_bench.log
python -m pytest tests/test_parallel_gpu.py -q -x --timeout 600 -s 2>&1 | tail -30 > gpurun_out/r02b/pytest_parallel.log
tail -12 gpurun_out/r02b/pytest_parallel.log
grep -h "ms_per_step" gpurun_out/r02b/*.json | cut -c1-400
grep "\[bench\]" gpurun_out/r02b/s2_bench.log | head -14
