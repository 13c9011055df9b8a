"""python scratch/kprobe_summary.py <dir with pass*/ subdirs>: per kernel (in launch order groups) average counters."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "conv_" not in k and "wgrad" not in k:
            continue
        key = (k, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        a = acc.setdefault(key, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
        if "End_Timestamp" in r:
            t = a.setdefault("dur_us", [0.0, 0])
            t[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; t[1] += 1
for key, a in acc.items():
    v = {k: c[0] / c[1] for k, c in a.items()}
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("\n", key)
    print("   dur_us %.1f  waves %.0f" % (v.get("dur_us", 0), v.get("SQ_WAVES", 0)))
    for k in ("SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
        if k in v: print("   %-22s %12.0f   /wave_cycles %.3f" % (k, v[k], v[k] / wc))
    for k in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA",
              "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "MfmaUtil"):
        if k in v: print("   %-28s %14.0f" % (k, v[k]))
