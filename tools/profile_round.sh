#!/bin/bash
# Regenerates what profiles/ holds for a round, on the GPU box (run through gpurun from the repo root):
#   the default bench line (headline + every leg of BASELINE.json's configurations from one run) with the per-shape log,
#   rocprofv3 kernel-trace summary of the stage-2 bench, and PMC passes (separate runs, kernel trace only, as the
#   pool requires): FETCH_SIZE, WRITE_SIZE (HBM-side traffic per kernel) and two SQ passes (MFMA busy / VALU).
# Outputs land in gpurun_out/$TAG; copy what is to be judged into profiles/ (tools/collect_profiles.sh).
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 --shapes > $O/s2_bench.json 2> $O/s2_bench.log        # (legs: s1, 32 views, fp8qk, fp8, VAE, simulated rank)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o s2 -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-profile --no-legs > $O/s2_bench_under_rocprof.log 2>&1
DB=$(ls /tmp/prof_kt/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/s2_kernel_stats.csv $O/s2_kernel_stats.txt 2> /dev/null
rm -rf /tmp/prof_kt
# round 4: the default command overlaps the two CFG halves of the large levels on two streams, so the kernel durations of
# the trace above are those of kernels SHARING the chip (their sum exceeds the step).  The same command with one stream
# (HI3D_TWO_STREAM=0: full-batch launches, nothing concurrent) gives the per-kernel durations free of overlap.
HI3D_TWO_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o s2 -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-profile --no-legs > $O/s2_bench_under_rocprof_one_stream.log 2>&1
DB=$(ls /tmp/prof_kt1/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $O/s2_kernel_stats_one_stream.csv $O/s2_kernel_stats_one_stream.txt 2> /dev/null
# (round 6: each .txt is written from its CSV's rows; a header over another file's rows -- profiles/r05_s2_kernel_stats.txt -- fails here)
for f in s2_kernel_stats s2_kernel_stats_one_stream; do
  python $R/tools/rocpd_summary.py --check $O/$f.txt $O/$f.csv || { echo "profile_round: $f.txt does not describe $f.csv" >&2; exit 1; }
done
rm -rf /tmp/prof_kt1
rocprofv3 -L > $O/rocprof_counters.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  HI3D_STEP_GRAPH=0 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-legs > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_traffic.py /tmp/pmc_$c $c > $O/s2_pmc_$c.csv
  rm -rf /tmp/pmc_$c
done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  HI3D_STEP_GRAPH=0 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_sq$i -o run --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-legs > $O/pmc_sq$i.log 2>&1
  python $R/tools/pmc_sum.py /tmp/pmc_sq$i > $O/s2_pmc_sq$i.csv 2>&1
  rm -rf /tmp/pmc_sq$i
done
# BASELINE config 5 (fp8 attention, both products): the same two SQ sets -- MFMA-busy against VALU-active of attn_d64_fp8_kernel
# ("for fp8 the P-pack / quantise VALU is now the bound -- show it in PMC")
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  HI3D_STEP_GRAPH=0 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_f8sq$i -o run --output-format csv -- python $R/bench.py --attn fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-legs > $O/pmc_fp8_sq$i.log 2>&1
  python $R/tools/pmc_sum.py /tmp/pmc_f8sq$i > $O/s2_fp8_pmc_sq$i.csv 2>&1
  rm -rf /tmp/pmc_f8sq$i
done
if [ -n "$LIGHT" ]; then     # (the per-shape conv traffic and the GEMM variant sweep only change when gemm.hip does)
  python $R/tools/make_traffic_json.py $O/s2_pmc_FETCH_SIZE.csv $O/s2_pmc_WRITE_SIZE.csv $O/traffic_s2.json
  cat $O/s2_bench.json | cut -c1-900; head -14 $O/s2_kernel_stats.csv; head -8 $O/s2_pmc_FETCH_SIZE.csv
  head -8 $O/s2_pmc_WRITE_SIZE.csv; head -14 $O/s2_pmc_sq1.csv; head -8 $O/s2_pmc_sq2.csv; grep -i "attn" $O/s2_fp8_pmc_sq1.csv $O/s2_fp8_pmc_sq2.csv
  exit 0
fi
# conv3x3 traffic per shape: FETCH_SIZE counts Infinity-Cache hits too, so the weights -- re-streamed
# from the 256 MB cache once per wave of m-tiles when K*N*2 exceeds the 4 MB L2 -- show up in it; the three levels separate
# the input halo (small weights, 128^2) from the weight stream (29.5 MB of weights, 32^2)
for shape in "32 128 320 320 res" "32 64 640 640 res" "32 32 1280 1280 res" "32 128 960 320"; do
  tag=$(echo $shape | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_cv -o run --output-format csv -- python $R/tools/kbench.py conv1 $shape > $O/pmc_conv_${tag}_$c.log 2>&1
    echo "== conv1 $shape  $c" >> $O/conv_traffic_per_shape.txt
    grep "^conv F" $O/pmc_conv_${tag}_$c.log >> $O/conv_traffic_per_shape.txt
    python $R/tools/pmc_traffic.py /tmp/pmc_cv $c | grep -i "gemm_bf16\|kernel," >> $O/conv_traffic_per_shape.txt
    rm -rf /tmp/pmc_cv
  done
done
cat $O/conv_traffic_per_shape.txt
cd $R; python tools/kbench.py sweep h 0 5 7 > $O/gemm_variant_sweep.log 2>&1; cd /tmp
grep "geglu" $O/gemm_variant_sweep.log
cat $O/s2_bench.json | cut -c1-700
head -12 $O/s2_kernel_stats.csv
head -8 $O/s2_pmc_FETCH_SIZE.csv
head -8 $O/s2_pmc_WRITE_SIZE.csv
head -14 $O/s2_pmc_sq1.csv; head -8 $O/s2_pmc_sq2.csv
