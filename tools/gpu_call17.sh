#!/bin/bash
# round 2, call 17: whole GPU suite on the new GEMM dispatch + stage-2 / stage-1 / VAE bench lines
O=gpurun_out/r02c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "not clip_parallel" 2>&1 | tail -15 > $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python bench.py --config s2 --steps 10 --warmup 3 --shapes --no-cpu-baseline > $O/s2_bench.json 2> $O/s2_bench.log
cut -c1-400 $O/s2_bench.json; head -13 $O/s2_bench.log
python bench.py --config s1 --steps 20 --warmup 3 --no-cpu-baseline > $O/s1_bench.json 2> $O/s1_bench.log
cut -c1-300 $O/s1_bench.json
python bench.py --config vae --steps 2 --warmup 1 > $O/vae_bench.json 2> $O/vae_bench.log
cut -c1-400 $O/vae_bench.json
