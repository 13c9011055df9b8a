set -e
O=gpurun_out/r02b; mkdir -p $O
for v in "" 1; do
for i in 1 2; do
PEMP_BN_TWO_LAUNCHES=$v timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --cpu-episodes 0 > $O/train_now.json 2> $O/train_now.err || { tail -20 $O/train_now.err; exit 1; }
python - "$v" <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/r02b/train_now.json") if l.startswith("{")][-1])
r=d["roofline"]
print("two_launches=%r" % sys.argv[1], d["ms_per_step"], {k:v["ms_per_step"] for k,v in r["by_class"].items()}, {k:v["ms_per_step"] for k,v in r["by_entry"].items() if "bn_" in k})
PY
done; done
