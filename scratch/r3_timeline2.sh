set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3tl; mkdir -p $O; rm -rf $O/kt
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --mode train $TRAIN_EXTRA --steps 20 --warmup 5 --cpu-episodes 0 --no-roofline > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
grep '^{' $O/kt.log | tail -n 1 | cut -c1-200
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 scratch/timeline.py $f $O/train_timeline.json 155 > $O/timeline.log 2>&1; head -70 $O/timeline.log
python3 scratch/sideq.py $f > $O/sideq.log 2>&1; tail -60 $O/sideq.log
rm -rf $O/kt
