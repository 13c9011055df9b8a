set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02b; mkdir -p $O
rm -rf $O/kt
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --mode train --steps 20 --warmup 5 --cpu-episodes 0 --no-e2e --no-single --no-roofline > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 scratch/profile_summary.py $f $O/train_steady.json train 155
python3 scratch/timeline.py $f $O/train_timeline.json 155 > /dev/null 2>&1; python3 scratch/tail_gap.py $f
rm -rf $O/kt
