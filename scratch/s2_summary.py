"""kernel_trace.csv of `rocprofv3 --kernel-trace` on `bench.py --model stage2 --shot 5 --batch 8` -> per-kernel averages over the
last <steps> steps: a step = the span between two launches of the stage-1 prior pass's first kernel (pack_input is launched
twice per step: prior pass, stage-2 pass).  python scratch/s2_summary.py trace.csv out.json <steps>"""
import csv, json, sys, collections
trace, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(trace))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
packs = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
per = 2
first = packs[-per * steps]
tail = rows[first:]
t0, t1 = int(tail[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in tail)
by = collections.OrderedDict()
busy = 0
for r in tail:
    a = by.setdefault(r["Kernel_Name"].split("(")[0], [0, 0])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[0] += 1
    a[1] += d
    busy += d
is_gemm = lambda k: "conv_dma" in k or "conv_igemm" in k
gemm_ns = sum(a[1] for k, a in by.items() if is_gemm(k))
rec = {"source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --model stage2 --shot 5 --batch 8 --steps 10 --warmup 3 "
                 "--cpu-episodes 0 --no-e2e --no-single --no-sides --no-roofline, PEMP_BENCH_LANES=1 (the kernel trace itself is not committed)",
       "note": f"steady state = the last {steps} steps (from the {per * steps}-th last pack_input launch on); one engine lane",
       "wall_ms_per_step": round((t1 - t0) / steps / 1e6, 4), "kernel_ms_per_step": round(busy / steps / 1e6, 4),
       "gemm_ms_per_step": round(gemm_ns / steps / 1e6, 4), "non_gemm_ms_per_step": round((busy - gemm_ns) / steps / 1e6, 4),
       "launches_per_step": round(len(tail) / steps, 1),
       "by_kernel": {k: {"launches_per_step": round(a[0] / steps, 2), "avg_us": round(a[1] / a[0] / 1e3, 2), "ms_per_step": round(a[1] / steps / 1e6, 4)}
                     for k, a in sorted(by.items(), key=lambda kv: -kv[1][1])}}
json.dump(rec, open(out, "w"), indent=1)
print(out, {k: rec[k] for k in ("wall_ms_per_step", "kernel_ms_per_step", "gemm_ms_per_step", "non_gemm_ms_per_step", "launches_per_step")})
