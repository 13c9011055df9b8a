set -e
O=gpurun_out/r02b; mkdir -p $O
for se in 1 2 4 13; do
PEMP_SEG_EVERY=$se timeout -k 10 300 python bench.py --mode train --train-graph --steps 40 --warmup 10 --no-single --no-roofline --cpu-episodes 0 > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - "$se" <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("graph seg_every", sys.argv[1], d["ms_per_step"], "host", d["config"].get("host_enqueue_ms_per_step"))
PY
done
