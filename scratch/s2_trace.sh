#!/bin/bash
# kernel trace of the stage-2 5-shot evaluation step (BASELINE.json configs[3]); summary -> gpurun_out/$R/stage2_5shot_steady.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${R:-r06}; O=gpurun_out/$R; mkdir -p $O
export PEMP_BENCH_LANES=1
rm -rf $O/kt_s2
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s2 -- python3 bench.py --model stage2 --shot 5 --batch 8 --steps 10 --warmup 3 --cpu-episodes 0 --no-e2e --no-single --no-sides --no-roofline > $O/kt_s2.log 2>&1 || echo "kt s2 failed"
find $O/kt_s2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/stage2_5shot_kernel_stats.csv
find $O/kt_s2 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scratch/s2_summary.py {} $O/stage2_5shot_steady.json 10
grep '^{' $O/kt_s2.log | tail -n 1 > $O/bench_stage2_5shot_under_rocprof.json
rm -rf $O/kt_s2
