"""kernel_trace.csv -> the kernel sequence of ONE steady-state step (the last complete one): name, duration, gap to the previous
kernel's end.  python scratch/step_sequence.py trace.csv <first-kernel-substring> [n_steps_back]"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"]); prev = t0
tot_k = tot_gap = 0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void pemp::", "").replace("pemp::", "")[:70]
    g = r.get("Grid_Size", r.get("Grid_Size_X", "")); wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))
    print(f"{(s - t0) / 1e3:8.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  grid {g:>8} wg {wg:>4}  {name}")
    tot_k += e - s; tot_gap += max(0, s - prev); prev = max(prev, e)
print(f"step: {len(seq)} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us wall, kernels {tot_k / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
