set -e
O=gpurun_out/r02b; mkdir -p $O
PEMP_BENCH_LAYERS=200 timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 10 --no-single --cpu-episodes 0 > $O/train_layers.json 2> $O/train_layers.err || { tail -20 $O/train_layers.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02b/train_layers.json") if l.startswith("{")][-1])
r=d["roofline"]; print(d["ms_per_step"], r["conv_ms_per_step"])
for l in r["by_layer"]:
    print(f'{l["kind"]:6s} M={l["M"]:6d} N={l["N"]:5d} K={l["K"]:5d} res={int(l["shortcut"])} share={l["share"]:.3f} ms={l["share"]*r["conv_ms_per_step"]:.3f} TF={l["tflops"]}')
PY
