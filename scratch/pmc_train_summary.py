"""rocprofv3 --pmc MfmaUtil pass of the training bench -> <round>_train_mfma_util.json (PEMP_ROUND, default r04) (per implicit-GEMM kernel, weighted by launch
duration).  python scratch/pmc_train_summary.py <mfma dir> <out dir>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import build
mdir, out = sys.argv[1], sys.argv[2]
f = glob.glob(mdir + "/**/*counter_collection.csv", recursive=True)[0]
by = collections.OrderedDict()
STEPS, PER_STEP = 5, 155          # the timed steps: the last 5 x 155 implicit-GEMM launches (autotune candidates come earlier)
gemm = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "MfmaUtil" and
        ("conv_dma" in r["Kernel_Name"] or "conv_wgrad" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"])]
gemm.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in gemm[-STEPS * PER_STEP:]:
    k = r["Kernel_Name"].split("(")[0]
    dur = float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0)) if "End_Timestamp" in r else 1.0
    dur = dur if dur > 0 else 1.0
    a = by.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"]) * dur; a[2] += dur
tw = sum(a[1] for a in by.values()); t = sum(a[2] for a in by.values())
cls = lambda k: "wgrad" if "wgrad" in k else "conv"
agg = {}
for k, a in by.items():
    c = agg.setdefault(cls(k), [0.0, 0.0]); c[0] += a[1]; c[1] += a[2]
json.dump({"command": "rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -- python3 bench.py --mode train --steps 5 --warmup 3 "
                      "--cpu-episodes 0 --no-single --no-roofline (counter collection serialises the two streams: per-kernel figures "
                      "are those of a kernel running alone; the last 5 x 155 implicit-GEMM launches = the timed steps)",
           "csrc_digest": build.csrc_digest(), "counter": "MfmaUtil, per launch, weighted by the launch's duration",
           "gemm_mfma_util_pct_time_weighted": round(tw / max(t, 1e-9), 2),
           "by_class": {c: round(v[0] / max(v[1], 1e-9), 1) for c, v in agg.items()},
           "by_kernel": {k: {"launches": a[0], "mfma_util_pct": round(a[1] / max(a[2], 1e-9), 1), "share_of_gemm_time": round(a[2] / max(t, 1e-9), 3)}
                         for k, a in sorted(by.items(), key=lambda kv: -kv[1][2])}},
          open(os.path.join(out, os.environ.get("PEMP_ROUND", "r04") + "_train_mfma_util.json"), "w"), indent=1)
print(open(os.path.join(out, os.environ.get("PEMP_ROUND", "r04") + "_train_mfma_util.json")).read()[:1500])
