set -e
mkdir -p gpurun_out/r02b
timeout -k 10 300 python -m pytest tests/test_train_ops_gpu.py -x -q -k "statistics" > gpurun_out/r02b/t_stats.log 2>&1 || { tail -30 gpurun_out/r02b/t_stats.log; exit 1; }
tail -2 gpurun_out/r02b/t_stats.log
PEMP_FUSE_BN_STATS=0 timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single > gpurun_out/r02b/train_nofuse.json 2> gpurun_out/r02b/train_nofuse.err
PEMP_FUSE_BN_STATS=1 timeout -k 10 300 python bench.py --mode train --steps 40 --warmup 10 --no-single > gpurun_out/r02b/train_fuse.json 2> gpurun_out/r02b/train_fuse.err
python - <<'PY'
import json
for n in ("nofuse","fuse"):
    d=json.loads([l for l in open(f"gpurun_out/r02b/train_{n}.json") if l.startswith("{")][-1])
    print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
PY
timeout -k 10 500 python -m pytest tests/test_train_gpu.py -x -q > gpurun_out/r02b/t_train.log 2>&1 || { tail -30 gpurun_out/r02b/t_train.log; exit 1; }
tail -2 gpurun_out/r02b/t_train.log
