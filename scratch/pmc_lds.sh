#!/bin/bash
# LDS counters of the eval and the training command (scratch/pmc_lds.py): R=r04 bash scratch/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${R:-r04}; export PEMP_ROUND=$R
O=gpurun_out/$R; mkdir -p $O
B="--cpu-episodes 0 --no-e2e --no-single --no-sides --no-roofline"
export PEMP_TILE_CACHE=$PWD/$O/tiles_lds.json PEMP_BENCH_LANES=1
timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 $B > /dev/null 2>&1
for c in LdsBankConflict LdsUtil; do
  rm -rf $O/lds_eval_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/lds_eval_$c -- python3 bench.py --steps 5 --warmup 2 $B > $O/lds_eval_$c.log 2>&1 || echo "eval $c failed"
done
python3 scratch/pmc_lds.py $O/lds_eval_LdsBankConflict $O/lds_eval_LdsUtil $O "eval --steps 5 --warmup 2" > /dev/null || echo "eval summary failed"
unset PEMP_TILE_CACHE PEMP_BENCH_LANES
for c in LdsBankConflict LdsUtil; do
  rm -rf $O/lds_train_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/lds_train_$c -- python3 bench.py --mode train --steps 5 --warmup 3 --cpu-episodes 0 --no-roofline > $O/lds_train_$c.log 2>&1 || echo "train $c failed"
done
python3 scratch/pmc_lds.py $O/lds_train_LdsBankConflict $O/lds_train_LdsUtil $O "train --mode train --steps 5 --warmup 3" > /dev/null || echo "train summary failed"
rm -rf $O/lds_eval_LdsBankConflict $O/lds_eval_LdsUtil $O/lds_train_LdsBankConflict $O/lds_train_LdsUtil
ls -la $O | grep lds
