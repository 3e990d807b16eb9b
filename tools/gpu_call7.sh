#!/bin/bash
# round 2, call 7: ping-pong wide tiles wired into the dispatch heuristic -- kernel + at-size parity, stage-2 / VAE bench, sweep
set -x
O=gpurun_out/r02b
mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_at_size_gpu.py -q -x --timeout 900 2>&1 | tail -8 > $O/pytest_kernels_atsize.log
tail -4 $O/pytest_kernels_atsize.log
python bench.py --config s2 --steps 10 --warmup 3 --shapes --no-cpu-baseline > $O/s2_bench.json 2> $O/s2_bench.log
cut -c1-1500 $O/s2_bench.json
head -14 $O/s2_bench.log
python bench.py --config vae --steps 2 --warmup 1 > $O/vae_bench.json 2> $O/vae_bench.log
cut -c1-600 $O/vae_bench.json
timeout 600 python tools/kbench.py sweep h 0 7 8 > $O/sweep2.log 2>&1
cat $O/sweep2.log
