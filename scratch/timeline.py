"""Timeline of the steady-state training steps from a rocprofv3 kernel trace: how busy the GPU is, how much of the
step two kernels run side by side, and where the main queue waits.  python scratch/timeline.py trace.csv out.json per_step"""
import csv, json, sys, collections
trace, out, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = 20
rows = [r for r in csv.DictReader(open(trace))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
is_gemm = lambda k: ("conv_dma" in k or "conv_igemm" in k or ("conv_wgrad" in k))
gemm = [r for r in rows if is_gemm(r["Kernel_Name"])][-steps * per_step:]
t0 = int(gemm[0]["Start_Timestamp"])
tail = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
t1 = max(int(r["End_Timestamp"]) for r in tail)
ev = []
for r in tail:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, busy, multi = 0, t0, 0, 0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
qkey = "Queue_Id" if "Queue_Id" in tail[0] else "Stream_Id"
queues = collections.defaultdict(list)
for r in tail: queues[r[qkey]].append(r)
qs = {}
for q, rs in queues.items():
    b = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = collections.Counter(); gap_ns = collections.Counter()
    for a, c in zip(rs, rs[1:]):
        g = int(c["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            k = c["Kernel_Name"].split("(")[0][:60]
            gaps[k] += 1; gap_ns[k] += g
    qs[q] = {"launches_per_step": round(len(rs) / steps, 1), "busy_ms_per_step": round(b / steps / 1e6, 3),
             "gap_before_ms_per_step": {k: round(v / steps / 1e6, 3) for k, v in gap_ns.most_common(8)},
             "total_gap_ms_per_step": round(sum(gap_ns.values()) / steps / 1e6, 3)}
rec = {"window_ms_per_step": round((t1 - t0) / steps / 1e6, 3), "busy_ms_per_step": round(busy / steps / 1e6, 3),
       "idle_ms_per_step": round((t1 - t0 - busy) / steps / 1e6, 3), "two_or_more_kernels_ms_per_step": round(multi / steps / 1e6, 3),
       "queues": qs}
# the largest idle intervals (no kernel on any queue) of the last step, with the kernels either side
last = [r for r in rows if is_gemm(r["Kernel_Name"])][-per_step:]
tl0 = int(last[0]["Start_Timestamp"])
lastrows = [r for r in rows if int(r["Start_Timestamp"]) >= tl0 - 3_000_000]
lastrows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = []
end = int(lastrows[0]["End_Timestamp"]); prev = lastrows[0]
for r in lastrows[1:]:
    st = int(r["Start_Timestamp"])
    if st > end:
        gaps.append((st - end, prev["Kernel_Name"].split("(")[0][:70], r["Kernel_Name"][:110]))
    if int(r["End_Timestamp"]) > end:
        end = int(r["End_Timestamp"]); prev = r
gaps.sort(reverse=True)
rec["largest_idle_intervals_us_last_step"] = [(round(g / 1e3, 1), a, b) for g, a, b in gaps[:25]]
rec["idle_total_us_last_window"] = round(sum(g for g, _, _ in gaps) / 1e3, 1)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
