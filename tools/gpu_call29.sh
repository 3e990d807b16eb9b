#!/bin/bash
# round 2 final: whole GPU suite (incl. the multi-process clip-parallel tests), then the profile round
O=gpurun_out/r02; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1200 2>&1 | tail -12 > $O/pytest_gpu_full.log
tail -5 $O/pytest_gpu_full.log
bash tools/profile_round.sh r02 2>&1 | tail -60
