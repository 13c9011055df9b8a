set -e
O=gpurun_out/r02b; mkdir -p $O
PEMP_BENCH_LAYERS=100 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-single --no-e2e --cpu-episodes 0 > $O/eval_layers.json 2> $O/eval_layers.err || { tail -20 $O/eval_layers.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02b/eval_layers.json") if l.startswith("{")][-1])
r=d["roofline"]; print(d["ms_per_step"], r["conv_ms_per_step"])
for l in r["by_layer"]:
    ms=l["share"]*r["conv_ms_per_step"]
    print(f'M={l["M"]:7d} N={l["N"]:5d} K={l["K"]:5d} res={int(l["shortcut"])} ms={ms:.3f} TF={l["tflops"]}')
PY
