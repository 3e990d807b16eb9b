#!/bin/bash
# The round's LAST GPU call: everything that is to be judged, produced from the final tree in one go on one
# box.  usage (from the build container):  gpurun --timeout 2400 -- "bash tools/final_round.sh <git HEAD> [tag]"
#   1. the full GPU suite, -x, with the tree's commit in the first line of the log      -> $O/pytest_gpu_final.log
#   2. __graft_entry__.smoke()                                                          -> $O/smoke.log
#   3. tools/profile_round.sh (LIGHT): driver-form bench line + per-shape log, rocprofv3 kernel traces (two streams / one
#      stream), PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ sets), traffic json          -> $O/s2_*, $O/traffic_s2.json
#   4. one stage-1 and one stage-2 clip end to end at full size                         -> $O/clip_e2e.json
#   5. bench.py's multi-GPU legs in a ONE-rank nccl group under torchrun (dry run of the code the driver runs on 2/4/8 GPUs)
# Copy what is to be judged from gpurun_out/$TAG into profiles/ afterwards (names: profiles/README.md, round-5 table).
HEAD=${1:-unknown}
TAG=${2:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
{
  echo "# git HEAD: $HEAD   (the tree this log was produced from; python -m pytest tests -m gpu -x -q)"
  python -m pytest tests -m gpu -x -q 2>&1 | tail -60
} > $O/pytest_gpu_final.log 2>&1
tail -3 $O/pytest_gpu_final.log
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
LIGHT=1 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-300
python tools/clip_e2e.py both > $O/clip_e2e.json 2> $O/clip_e2e.log; cut -c1-600 $O/clip_e2e.json
HI3D_BENCH_FORCE_MULTI=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/bench_multi_gpu_legs_dry_run.json 2> $O/bench_multi_gpu_legs_dry_run.log
python - <<EOF
import json
d = json.load(open("$O/bench_multi_gpu_legs_dry_run.json"))
for k in ("clip_parallel", "clip_parallel_32views", "vae_decode_sharded", "clip_parallel_cfg1_overlap"):
    print(k, json.dumps(d.get(k))[:260])
EOF
# 6. the opt-in 4-process clip-parallel case (cfg 2 x sp 2 on ONE GPU, four processes time-slicing it: ~4 min)
HI3D_SLOW_TESTS=1 timeout 900 python -m pytest tests/test_parallel_gpu.py -m gpu -q -k "test_clip_parallel_step_matches_single_gpu and 4-2" 2>&1 | tail -6 > $O/pytest_slow_4proc.log; tail -2 $O/pytest_slow_4proc.log
