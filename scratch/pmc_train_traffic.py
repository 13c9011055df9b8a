"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the training bench -> <round>_train_traffic.json (PEMP_ROUND, default r04): HBM bytes per step, by kernel class
(the last 5 steps: everything from the first of the last 5 x 155 implicit-GEMM launches on).
python scratch/pmc_train_traffic.py <fetch dir> <write dir> <out dir>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import build
fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
STEPS, PER_STEP = 5, 155
is_gemm = lambda k: "conv_dma" in k or "conv_wgrad" in k or "conv_igemm" in k


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    g = [i for i, r in enumerate(rows) if is_gemm(r["Kernel_Name"])]
    return rows[g[-STEPS * PER_STEP]:]


def cls(k):
    if "wgrad_reduce" in k: return "wgrad reduce"
    if "conv_wgrad" in k: return "wgrad"
    if is_gemm(k): return "conv"
    if "bn_" in k or "colsum" in k: return "batchnorm"
    if "at::native" in k or "rocclr" in k: return "torch glue"
    return "other"


tot = collections.OrderedDict()
for d, c, mul in ((fdir, "FETCH_SIZE", 2.0), (wdir, "WRITE_SIZE", 1.0)):      # gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)
    for r in load(d, c):
        k = cls(r["Kernel_Name"])
        t = tot.setdefault(k, {"read_MB": 0.0, "write_MB": 0.0})
        t["read_MB" if c == "FETCH_SIZE" else "write_MB"] += float(r["Counter_Value"]) * mul / 1024 / STEPS
rec = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py --mode train "
                  "--steps 5 --warmup 3 --cpu-episodes 0 --no-single --no-roofline; the last 5 steps",
       "csrc_digest": build.csrc_digest(),
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2; WRITE_SIZE exact",
       "hbm_MB_per_step": {k: {a: round(b, 1) for a, b in v.items()} for k, v in tot.items()},
       "hbm_GB_per_step_total": round(sum(v["read_MB"] + v["write_MB"] for v in tot.values()) / 1024, 3)}
# what bench.py quotes as `train.roofline.traffic`: HBM bytes per implicit-GEMM launch (convs, weight gradients and their second pass)
gemm_mb = sum(tot[k]["read_MB"] + tot[k]["write_MB"] for k in ("conv", "wgrad", "wgrad reduce") if k in tot)
rec["gemm_launches_per_step"] = PER_STEP
rec["gemm_hbm_bytes_per_launch"] = int(gemm_mb * 1024 * 1024 / PER_STEP)
json.dump(rec, open(os.path.join(out, os.environ.get("PEMP_ROUND", "r04") + "_train_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
