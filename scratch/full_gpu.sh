set -e
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --cpu-episodes 0 > $O/train_now.json 2> $O/train_now.err || { tail -20 $O/train_now.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02b/train_now.json") if l.startswith("{")][-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "host", d["config"]["host_enqueue_ms_per_step"], {k:v["ms_per_step"] for k,v in r["by_class"].items()})
print("   ", {k:v["ms_per_step"] for k,v in r["by_entry"].items()})
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -3 $O/t_all.log
