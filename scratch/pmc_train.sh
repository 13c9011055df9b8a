set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
rm -rf $O/pmc_train
timeout -k 10 600 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/pmc_train -- python3 bench.py --mode train --steps 5 --warmup 3 --cpu-episodes 0 --no-single --no-roofline > $O/pmc_train.log 2>&1 || { tail $O/pmc_train.log; exit 1; }
f=$(find $O/pmc_train -name "*counter_collection.csv" | head -1); head -1 $f
python3 scratch/pmc_train_summary.py $O/pmc_train $O
rm -rf $O/pmc_train
