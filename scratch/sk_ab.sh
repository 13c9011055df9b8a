set -e
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_train_ops_gpu.py -x -q -k "statistics or epilogue" > $O/t_ops.log 2>&1 || { tail -40 $O/t_ops.log; exit 1; }
tail -2 $O/t_ops.log
for f in 0 1; do
PEMP_CONV_SPLITK=$f timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --cpu-episodes 0 > $O/train_sk$f.json 2> $O/train_sk$f.err || { tail -20 $O/train_sk$f.err; exit 1; }
done
python - <<'PY'
import json
for n in ("0","1"):
    d=json.loads([l for l in open(f"gpurun_out/r02b/train_sk{n}.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("splitk",n, d["value"], d["ms_per_step"], "host", d["config"]["host_enqueue_ms_per_step"], {k:v["ms_per_step"] for k,v in r["by_class"].items()})
    print("   ", {k:v["ms_per_step"] for k,v in r["by_entry"].items()})
    for l in r["by_layer"]: print("   ", l)
PY
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q > $O/t_train.log 2>&1 || { tail -40 $O/t_train.log; exit 1; }
tail -2 $O/t_train.log
