set -e
O=gpurun_out/r02b; mkdir -p $O
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for pr in 0 -1 1; do
for g in "" "--train-graph"; do
PEMP_SIDE_PRIORITY=$pr timeout -k 10 300 python bench.py --mode train $g --steps 40 --warmup 10 --no-single --no-roofline --cpu-episodes 0 > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - "$pr" "$g" <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("side prio", sys.argv[1], sys.argv[2] or "eager", d["ms_per_step"], "host", d["config"].get("host_enqueue_ms_per_step"))
PY
done; done
