#!/bin/bash
# Regenerates what profiles/ holds, on the GPU box (run through gpurun from the repo root):
#   bench logs (stage 2 with CPU baseline, stage 1), rocprofv3 kernel trace summary, and the two PMC
#   passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel trace only, as the pool requires).
# Outputs land in gpurun_out/final_*; copy them to profiles/ and rebuild profiles/traffic_s2.json with
#   python tools/make_traffic_json.py gpurun_out/final_traffic_FETCH_SIZE.csv gpurun_out/final_traffic_WRITE_SIZE.csv profiles/traffic_s2.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --config s2 > $O/final_s2_bench.log 2>&1
python bench.py --config s1 --no-cpu-baseline > $O/final_s1_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_r01c
rocprofv3 --kernel-trace --stats -d $O/prof_r01c -o s2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/final_s2_bench_under_rocprof.log 2>&1
DB=$(ls $O/prof_r01c/*.db | head -1)
python $R/tools/rocpd_summary.py $DB $O/final_s2_kernel_stats.csv 2> $O/final_s2_kernel_stats.txt
rm -rf $O/prof_r01c
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/final_pmc_$c.log 2>&1
  python $R/tools/pmc_traffic.py /tmp/pmc_$c $c > $O/final_traffic_$c.csv
  rm -rf /tmp/pmc_$c
done
tail -1 $O/final_s2_bench.log | cut -c1-600
tail -1 $O/final_s1_bench.log | cut -c1-300
head -8 $O/final_s2_kernel_stats.csv
head -6 $O/final_traffic_FETCH_SIZE.csv
head -6 $O/final_traffic_WRITE_SIZE.csv
