set -e
O=gpurun_out/r02b; mkdir -p $O
for i in 1 2; do
timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --no-roofline --cpu-episodes 0 > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("train", d["ms_per_step"])
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-single --no-e2e --cpu-episodes 0 > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("eval", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
