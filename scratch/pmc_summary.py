"""Three rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil; --kernel-trace only, CSV) of the eval bench ->
<round>_conv_traffic.json / <round>_mfma_util.json (PEMP_ROUND, default r04) with the kernel-source digest bench.py checks before quoting them.
python scratch/pmc_summary.py <fetch dir> <write dir> <mfma dir> <out dir> <episodes per step>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import build
fdir, wdir, mdir, out, batch = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
is_conv = lambda k: ("conv_dma" in k or "conv_igemm" in k)


def rows(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]


def avg(d, counter):
    v = [float(r["Counter_Value"]) for r in rows(d, counter) if is_conv(r["Kernel_Name"])]
    return sum(v) / len(v), len(v)


digest = build.conv_digest()
RND = os.environ.get("PEMP_ROUND", "r04")
key = f"stage1-eval-b{batch}-s1"
fk, n = avg(fdir, "FETCH_SIZE")
wk, _ = avg(wdir, "WRITE_SIZE")
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py --steps 5 "
                      "--warmup 2 --cpu-episodes 0 --no-e2e --no-single --no-roofline (%d episodes/step; tile picks replayed through PEMP_TILE_CACHE)" % batch,
           "conv_digest": digest, "workload_key": key, "episodes_per_step": batch, "conv_launches": n,
           "FETCH_SIZE_KB_avg_per_launch": round(fk, 2), "WRITE_SIZE_KB_avg_per_launch": round(wk, 2),
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}, open(os.path.join(out, RND + "_conv_traffic.json"), "w"), indent=1)
# MfmaUtil: time-weight by the kernel's duration (same CSV: Start/End timestamps per dispatch)
by = collections.OrderedDict()
tot_w = tot = 0.0
cosine = []
for r in rows(mdir, "MfmaUtil"):
    k = r["Kernel_Name"].split("(")[0]
    dur = float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0)) if "End_Timestamp" in r else 1.0
    if dur <= 0:
        dur = 1.0
    if is_conv(k):
        a = by.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"]) * dur; a[2] += dur
        tot_w += float(r["Counter_Value"]) * dur; tot += dur
    elif "cosine_mfma" in k:
        cosine.append(float(r["Counter_Value"]))
json.dump({"command": "rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-episodes 0 --no-e2e "
                      "--no-single --no-roofline (%d episodes/step; tile picks replayed through PEMP_TILE_CACHE)" % batch,
           "counter": "MfmaUtil (rocprofv3 derived counter: MFMA pipe busy share), per launch, weighted by the launch's duration",
           "conv_digest": digest, "workload_key": key, "episodes_per_step": batch, "conv_launches": sum(a[0] for a in by.values()),
           "conv_mfma_util_pct_time_weighted": round(tot_w / max(tot, 1e-9), 2),
           "by_kernel": {k: {"launches": a[0], "mfma_util_pct": round(a[1] / max(a[2], 1e-9), 1)} for k, a in by.items()},
           "cosine_mfma_kernel_mfma_util_pct": round(sum(cosine) / max(len(cosine), 1), 1)},
          open(os.path.join(out, RND + "_mfma_util.json"), "w"), indent=1)
print(open(os.path.join(out, RND + "_conv_traffic.json")).read()); print(open(os.path.join(out, RND + "_mfma_util.json")).read())
