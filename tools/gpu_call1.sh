#!/bin/bash
# round-2 GPU call 1: full GPU test suite, stage-2 / stage-1 / VAE bench with per-shape tables, conv K-order A/B
set -x
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -40 > gpurun_out/r02/pytest_gpu.log
python bench.py --steps 10 --warmup 3 --shapes > gpurun_out/r02/s2_bench.json 2> gpurun_out/r02/s2_bench.log
HI3D_CONV_KORDER=0 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r02/s2_korder0.json 2> gpurun_out/r02/s2_korder0.log
HI3D_STEP_GRAPH=0 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/r02/s2_nograph.json 2> gpurun_out/r02/s2_nograph.log
python bench.py --config s1 --steps 20 --warmup 3 --shapes --no-cpu-baseline > gpurun_out/r02/s1_bench.json 2> gpurun_out/r02/s1_bench.log
HI3D_STEP_GRAPH=0 python bench.py --config s1 --steps 20 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/r02/s1_nograph.json 2> gpurun_out/r02/s1_nograph.log
python bench.py --config vae --steps 2 --warmup 1 > gpurun_out/r02/vae_bench.json 2> gpurun_out/r02/vae_bench.log
tail -5 gpurun_out/r02/pytest_gpu.log
cat gpurun_out/r02/*.json
