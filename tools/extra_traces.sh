#!/bin/bash
# Extra evidence of a round (run through gpurun from the repo root): rocprofv3 kernel-trace summaries of the 32-view leg (BASELINE
# config 4 on one GPU: shows attn_temporal_mfma32_kernel) and of stage 1 (config 2), and one more default `python bench.py` line
# (another box of the pool: the headline's box-to-box spread).   usage: bash tools/extra_traces.sh [tag]
TAG=${1:-r06x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cfg in "s2_32views --views 32" "s1 --config s1"; do
  set -- $cfg; name=$1; shift
  rm -rf /tmp/prof_x
  rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o t -- python $R/bench.py "$@" --steps 4 --warmup 3 --no-cpu-baseline --no-profile --no-legs > $O/${name}_bench_under_rocprof.log 2>&1
  DB=$(ls /tmp/prof_x/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/${name}_kernel_stats.csv $O/${name}_kernel_stats.txt 2> /dev/null
  head -8 $O/${name}_kernel_stats.csv | cut -c1-150
done
rm -rf /tmp/prof_x
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], {k:(v.get('ms_per_step') or v.get('rank_ms_per_step')) for k,v in d['legs'].items()})
"
