"""rocprofv3 --pmc LdsBankConflict / LdsUtil passes (separate, --kernel-trace only, CSV) of a bench command -> <round>_lds_<tag>.json:
per implicit-GEMM kernel, weighted by the launch's duration.  python scratch/pmc_lds.py <conflict dir> <util dir> <out dir> <tag>"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import build
cdir, udir, out, tag = sys.argv[1:5]
is_gemm = lambda k: ("conv_dma" in k or "conv_igemm" in k or "conv_wgrad" in k) and "reduce" not in k


def table(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or not is_gemm(r["Kernel_Name"]):
            continue
        k = r["Kernel_Name"].split("(")[0]
        dur = max(float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0)), 1.0)
        a = by.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"]) * dur; a[2] += dur
    tw = sum(a[1] for a in by.values()) / max(sum(a[2] for a in by.values()), 1e-9)
    return by, tw


cf, cf_all = table(cdir, "LdsBankConflict")
ut, ut_all = table(udir, "LdsUtil")
rec = {"command": "rocprofv3 --pmc LdsBankConflict | LdsUtil (separate passes) --kernel-trace --output-format csv -- python3 bench.py %s" % tag.split(" ", 1)[1],
       "csrc_digest": build.csrc_digest(),
       "counters": "LdsBankConflict = SQ_LDS_BANK_CONFLICT / (SQ_LDS_IDX_ACTIVE - SQ_LDS_BANK_CONFLICT): cycles lost to bank conflicts per useful LDS "
                   "cycle; LdsUtil = 100 * SQ_LDS_IDX_ACTIVE / (busy cycles * CUs): share of the time the LDS of a CU is executing indexed "
                   "operations (ds_read / ds_write; the LDS-DMA fills are not indexed operations); per launch, weighted by its duration",
       "gemm_lds_bank_conflict_ratio_time_weighted": round(cf_all, 4), "gemm_lds_util_pct_time_weighted": round(ut_all, 1),
       "by_kernel": {k: {"launches": a[0], "bank_conflict_ratio": round(a[1] / max(a[2], 1e-9), 4),
                         "lds_util_pct": round(ut[k][1] / max(ut[k][2], 1e-9), 1) if k in ut else None} for k, a in cf.items()}}
name = os.path.join(out, os.environ.get("PEMP_ROUND", "r04") + "_lds_" + tag.split()[0].strip("-") + ".json")
json.dump(rec, open(name, "w"), indent=1)
print(json.dumps(rec, indent=1)[:3000])
