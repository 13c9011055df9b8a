set -e
O=gpurun_out/r02b; mkdir -p $O
for v in 1 0 1 0; do
PEMP_WGRAD_XCD=$v timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --no-roofline --cpu-episodes 0 > $O/p.json 2> $O/p.err || { tail -5 $O/p.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/p.json") if l.startswith("{")][-1])
print("wgrad xcd", sys.argv[1], d["ms_per_step"])
PY
done
