"""kernel_trace.csv of `rocprofv3 --kernel-trace` on bench.py -> steady-state per-kernel averages: the LAST
20 x <conv launches per step> implicit-GEMM launches of the trace (= the 20 timed steps; autotune candidates and warm-up
come earlier).  python scratch/profile_summary.py trace.csv out.json eval|train <conv launches per step>"""
import csv, json, sys, collections
trace, out, tag, per_step = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
steps = 20
rows = [r for r in csv.DictReader(open(trace))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_gemm = lambda k: ("conv_dma" in k or "conv_igemm" in k or ("conv_wgrad" in k))
gemm = [r for r in rows if is_gemm(r["Kernel_Name"])][-steps * per_step:]
t_first = int(gemm[0]["Start_Timestamp"])
tail = [r for r in rows if int(r["Start_Timestamp"]) >= t_first]
by = collections.OrderedDict()
for r in tail:
    a = by.setdefault(r["Kernel_Name"].split("(")[0], [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
conv_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in gemm)
rec = {"source": f"rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py {'--mode train ' if tag == 'train' else '--loss cedt ' if tag == 'evalcedt' else ''}--steps 20 --warmup 5 "
                 "--cpu-episodes 0 --no-e2e --no-single --no-roofline (the kernel trace itself is not committed)",
       "note": f"steady state = everything from the first of the last {steps} x {per_step} implicit-GEMM launches on (the {steps} timed steps)",
       "gemm_launches": len(gemm), "gemm_avg_launch_us": round(conv_ns / len(gemm) / 1e3, 2), "gemm_ms_per_step": round(conv_ns / steps / 1e6, 4),
       "by_kernel": {k: {"launches_per_step": round(a[0] / steps, 2), "avg_us": round(a[1] / a[0] / 1e3, 2), "ms_per_step": round(a[1] / steps / 1e6, 4)}
                     for k, a in sorted(by.items(), key=lambda kv: -kv[1][1])}}
json.dump(rec, open(out, "w"), indent=1)
print(out, "gemm avg launch us", rec["gemm_avg_launch_us"], "gemm ms/step", rec["gemm_ms_per_step"])
