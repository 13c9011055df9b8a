"""Last training step of a kernel trace: per queue, when its first / last kernel of the backward pass ran, and for the side queue
(weight gradients) how far behind the main queue it finishes.  python scratch/sideq.py trace.csv"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
opt = [i for i, r in enumerate(rows) if "sgd_clip" in r["Kernel_Name"]]
a, b = opt[-2], opt[-1]
st = rows[a + 1:b + 1]
t0 = rows[a]["e"]
qkey = "Queue_Id" if "Queue_Id" in st[0] else "Stream_Id"
qs = collections.defaultdict(list)
for r in st:
    qs[r[qkey]].append(r)
for q, rs in qs.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, first start {(rs[0]['s'] - t0) / 1e3:.1f} us, last end {(max(r['e'] for r in rs) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
    wg = [r for r in rs if "wgrad" in r["Kernel_Name"]]
    if wg:
        print("   wgrad kernels: first start %.1f us, last end %.1f us" % ((wg[0]["s"] - t0) / 1e3, (max(r["e"] for r in wg) - t0) / 1e3))
main = max(qs.values(), key=len)
side = [rs for rs in qs.values() if rs is not main and any("wgrad" in r["Kernel_Name"] for r in rs)]
print("step length %.1f us" % ((st[-1]["e"] - t0) / 1e3))
# main queue kernels after the side queue's last wgrad ended, and vice versa
if side:
    s_end = max(r["e"] for r in side[0])
    m_last_gemm = max(r["e"] for r in main if "conv_dma" in r["Kernel_Name"])
    print("main queue's last conv ends at %.1f us, side queue ends at %.1f us" % ((m_last_gemm - t0) / 1e3, (s_end - t0) / 1e3))
    # timeline in 0.5 ms bins: main gemm busy, side busy
    T = st[-1]["e"] - t0
    nb = int(T / 5e5) + 1
    mb, sb, ob = [0.0] * nb, [0.0] * nb, [0.0] * nb
    def add(arr, r):
        s, e = r["s"] - t0, r["e"] - t0
        for k in range(int(s / 5e5), min(int(e / 5e5), nb - 1) + 1):
            lo, hi = max(s, k * 5e5), min(e, (k + 1) * 5e5)
            if hi > lo: arr[k] += (hi - lo) / 5e5
    for r in main:
        add(mb if ("conv_dma" in r["Kernel_Name"] or "conv_igemm" in r["Kernel_Name"]) else ob, r)
    for r in side[0]:
        add(sb, r)
    print("bin(0.5ms)  main-gemm  main-other  side")
    for k in range(nb):
        print("%5.1f ms   %5.2f      %5.2f      %5.2f" % (k * 0.5, mb[k], ob[k], sb[k]))
