set -e
O=gpurun_out/r02b; mkdir -p $O
for l in 4 6 8; do
PEMP_EVAL_LANES=$l timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-e2e --cpu-episodes 0 --no-roofline > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - $l <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("lanes", sys.argv[1], d["single_episode"])
PY
done
