set -e
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_conv_fuzz_gpu.py tests/test_train_ops_gpu.py -x -q > $O/t_w.log 2>&1 || { tail -40 $O/t_w.log; exit 1; }
tail -2 $O/t_w.log
for i in 1 2; do
timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single --cpu-episodes 0 > $O/train_now.json 2> $O/train_now.err || { tail -20 $O/train_now.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02b/train_now.json") if l.startswith("{")][-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "host", d["config"]["host_enqueue_ms_per_step"], {k:v["ms_per_step"] for k,v in r["by_class"].items()}, r.get("step_effective_tflops"))
PY
done
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q > $O/t_train.log 2>&1 || { tail -40 $O/t_train.log; exit 1; }
tail -2 $O/t_train.log
