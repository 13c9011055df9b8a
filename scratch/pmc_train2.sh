set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmct_$c
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmct_$c -- python3 bench.py --mode train --steps 5 --warmup 3 --cpu-episodes 0 --no-single --no-roofline > $O/pmct_$c.log 2>&1 || { tail $O/pmct_$c.log; exit 1; }
done
python3 scratch/pmc_train_traffic.py $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE $O
rm -rf $O/pmct_FETCH_SIZE $O/pmct_WRITE_SIZE
