#!/bin/bash
# Produces the round's measurement artefacts under gpurun_out/$R/ on the GPU box (copy into profiles/ afterwards, prefixed with $R_):
#   [R=r05] bash scratch/make_profiles.sh [part]        part = bench | trace | b1 | head | pmc | pmctrain | all (default)
# rocprofv3 is always given the program itself after `--` (python3 bench.py ...), never a wrapper.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${1:-all}
R=${R:-r06}
export PEMP_ROUND=$R
O=gpurun_out/$R; mkdir -p $O
B="--cpu-episodes 0 --no-e2e --no-single --no-sides"
if [ $PART = bench ] || [ $PART = all ]; then
# 1. bench lines: the default command (headline + comm / cedt / protocol_5x1000 / train / stage2_5shot / miou objects), then every other
#    BASELINE.json config
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/err.log && echo default ok
timeout -k 10 500 python3 bench.py --mode train --steps 20 --warmup 5 > $O/bench_train_b4.json 2>> $O/err.log && echo train ok
timeout -k 10 500 python3 bench.py --mode train --loss cedt --steps 20 --warmup 5 --cpu-episodes 0 > $O/bench_train_cedt_b4.json 2>> $O/err.log && echo train cedt ok
timeout -k 10 500 python3 bench.py --loss cedt --steps 20 --warmup 5 --cpu-episodes 0 > $O/bench_eval_cedt_b25.json 2>> $O/err.log && echo eval cedt ok
timeout -k 10 500 python3 bench.py --mode train --model stage2 --steps 20 --warmup 5 > $O/bench_train_stage2_b4.json 2>> $O/err.log && echo train2 ok
timeout -k 10 500 python3 bench.py --model stage2 --shot 5 --batch 8 --steps 10 --warmup 3 > $O/bench_stage2_5shot_b8.json 2>> $O/err.log && echo stage2 ok
timeout -k 10 500 python3 bench.py --dataset COCO --steps 10 --warmup 3 --cpu-episodes 4 > $O/bench_eval_coco_b25.json 2>> $O/err.log && echo coco ok
timeout -k 10 500 python3 bench.py --model baseline --batch 12 --steps 10 --warmup 3 > $O/bench_baseline_vgg16_b12.json 2>> $O/err.log && echo baseline ok
PEMP_EVAL_SPLITK=1 timeout -k 10 500 python3 bench.py --batch 1 --steps 100 --warmup 10 $B > $O/bench_eval_b1.json 2>> $O/err.log && echo b1 ok
timeout -k 10 500 python3 bench.py --batch 1 --steps 100 --warmup 10 $B > $O/bench_eval_b1_exact.json 2>> $O/err.log && echo b1 exact ok
fi
if [ $PART = trace ] || [ $PART = all ]; then
# 2. kernel traces (rocprofv3 --kernel-trace --stats) of the eval and train commands.  The eval command runs with ONE engine lane
#    here (PEMP_BENCH_LANES=1; exported, not passed through a wrapper): with the default two lanes the kernels of consecutive steps
#    overlap and a trace's per-kernel durations are those of kernels sharing the chip, not of the kernel -- bench.py's live
#    roofline pass (single lane, events per launch) is what these averages are compared with.
export PEMP_BENCH_LANES=1
for tag in eval train evalcedt; do
  extra=""; [ $tag = train ] && extra="--mode train"; [ $tag = evalcedt ] && extra="--loss cedt"
  rm -rf $O/kt_$tag
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 bench.py $extra --steps 20 --warmup 5 $B --no-roofline > $O/kt_$tag.log 2>&1 || echo "kt $tag failed"
  find $O/kt_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${tag}_kernel_stats.csv
  per=52; [ $tag = train ] && per=155
  find $O/kt_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/profile_summary.py {} $O/${tag}_steady.json $tag $per
  [ $tag = train ] && find $O/kt_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/phases3.py {} 20 > $O/train_phases.json
  grep '^{' $O/kt_$tag.log | tail -n 1 > $O/bench_${tag}_under_rocprof.json
  rm -rf $O/kt_$tag
done
unset PEMP_BENCH_LANES
fi
if [ $PART = b1 ] || [ $PART = all ]; then
# 2b. the one-episode step (split-K variants allowed, one lane) kernel by kernel
export PEMP_BENCH_LANES=1
rm -rf $O/kt_b1x
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b1x -- python3 bench.py --batch 1 --steps 20 --warmup 5 $B --no-roofline > $O/kt_b1x.log 2>&1 || echo "kt b1 exact failed"
find $O/kt_b1x -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/profile_summary.py {} $O/eval_b1_exact_steady.json eval 46
find $O/kt_b1x -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/step_sequence.py {} pack_input > $O/eval_b1_exact_sequence.txt
rm -rf $O/kt_b1x
export PEMP_EVAL_SPLITK=1
rm -rf $O/kt_b1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b1 -- python3 bench.py --batch 1 --steps 20 --warmup 5 $B --no-roofline > $O/kt_b1.log 2>&1 || echo "kt b1 failed"
find $O/kt_b1 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/profile_summary.py {} $O/eval_b1_steady.json eval 46
rm -rf $O/kt_b1
unset PEMP_BENCH_LANES PEMP_EVAL_SPLITK
fi
if [ $PART = head ] || [ $PART = all ]; then
# 4. prototype-head kernels alone at 133 / 320 / 533 MB per launch (the last two beyond the 256 MB Infinity Cache); CELossDT kernels
for b in 25 60 100; do
  rm -rf $O/kt_head
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_head -- python3 scratch/head_bench.py $b > $O/head_b$b.log 2>&1 || echo "head $b failed"
  find $O/kt_head -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/head_b${b}_kernel_stats.csv
done
rm -rf $O/kt_head
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cedt -- python3 scratch/cedt_bench.py > $O/cedt.log 2>&1 || echo "cedt failed"
find $O/kt_cedt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cedt_kernel_stats.csv
rm -rf $O/kt_cedt
fi
if [ $PART = pmc ] || [ $PART = all ]; then
# 3. PMC passes on the eval command (tile picks replayed, so no autotune launches): HBM traffic + MFMA utilisation
export PEMP_TILE_CACHE=$PWD/$O/tiles.json
export PEMP_BENCH_LANES=1
timeout -k 10 400 python3 bench.py --steps 3 --warmup 2 $B --no-roofline > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rm -rf $O/pmc_$c
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 5 --warmup 2 $B --no-roofline > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
python3 scratch/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil $O 25 > /dev/null || echo "pmc summary failed"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil
unset PEMP_TILE_CACHE PEMP_BENCH_LANES
fi
if [ $PART = pmctrain ] || [ $PART = all ]; then
# 3b. MfmaUtil and HBM traffic of the training step
rm -rf $O/pmc_train
timeout -k 10 600 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pmc_train -- python3 bench.py --mode train --steps 5 --warmup 3 --cpu-episodes 0 --no-roofline > $O/pmc_train.log 2>&1 || echo "pmc train failed"
python3 scratch/pmc_train_summary.py $O/pmc_train $O > /dev/null || echo "pmc train summary failed"
rm -rf $O/pmc_train
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmct_$c
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmct_$c -- python3 bench.py --mode train --steps 5 --warmup 3 --cpu-episodes 0 --no-roofline > $O/pmct_$c.log 2>&1 || echo "pmc train $c failed"
done
python3 scratch/pmc_train_traffic.py $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE $O > /dev/null || echo "pmc train traffic failed"
rm -rf $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE
fi
rm -f $O/*.log.tmp
ls -la $O
