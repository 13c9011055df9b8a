"""Training-step timeline by phase from a rocprofv3 kernel trace: for forward / backward / optimizer, the wall time, the time
with at least one implicit-GEMM kernel running, the time with only other kernels running, and the idle time.
python scratch/phases3.py trace.csv [steps]"""
import csv, sys, json, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
is_gemm = lambda k: ("conv_dma" in k or "conv_igemm" in k or "conv_wgrad" in k)
opt = [i for i, r in enumerate(rows) if "sgd_clip" in r["Kernel_Name"]]
opt = opt[-(steps + 1):]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
only_other = collections.Counter()
for a, b in zip(opt, opt[1:]):
    st = rows[a + 1:b + 1]
    t0 = rows[a]["e"]
    # phase boundaries: first backward kernel of the head, the optimizer's first kernel (sqsum)
    tb = next((r["s"] for r in st if "upsample_bwd" in r["Kernel_Name"] or "head_bwd" in r["Kernel_Name"] or "cosine_bwd" in r["Kernel_Name"]), None)
    to = next((r["s"] for r in st if "sqsum" in r["Kernel_Name"]), st[-1]["s"])
    t1 = st[-1]["e"]
    bounds = [("forward", t0, tb), ("backward", tb, to), ("optimizer", to, t1)]
    ev = []
    for r in st:
        g = is_gemm(r["Kernel_Name"])
        ev.append((r["s"], 1, g, r["Kernel_Name"])); ev.append((r["e"], -1, g, r["Kernel_Name"]))
    ev.sort(key=lambda x: (x[0], x[1]))
    for name, p0, p1 in bounds:
        ng = no = 0
        last = p0
        running = collections.Counter()
        for t, d, g, k in ev:
            tt = min(max(t, p0), p1)
            if tt > last:
                dt = tt - last
                if ng > 0: acc[name]["gemm"] += dt
                elif no > 0:
                    acc[name]["other_only"] += dt
                    for kk in running: only_other[(name, kk.split("(")[0][:50])] += dt / max(len(running), 1)
                else: acc[name]["idle"] += dt
                if ng > 0 and no > 0: acc[name]["gemm_and_other"] += dt
                if ng > 1: acc[name]["two_gemms"] += dt
                last = tt
            if g: ng += d
            else:
                no += d
                if d > 0: running[k] += 1
                else:
                    running[k] -= 1
                    if running[k] <= 0: del running[k]
        acc[name]["wall"] += p1 - p0
out = {p: {k: round(v / steps / 1e6, 3) for k, v in d.items()} for p, d in acc.items()}
out["step_ms"] = round(sum(d["wall"] for d in acc.values()) / steps / 1e6, 3)
out["other_only_top_ms_per_step"] = {f"{p}:{k}": round(v / steps / 1e6, 3) for (p, k), v in only_other.most_common(25)}
print(json.dumps(out, indent=1))
