"""How long does the main queue wait for the side queue at the end of the backward pass (before the optimizer)?"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sqsum_partial_kernel" in r["Kernel_Name"]]
for i in idx[-6:]:
    s = int(rows[i]["Start_Timestamp"])
    q = rows[i]["Queue_Id"]
    # last kernel on the same queue before it
    j = i - 1
    while rows[j]["Queue_Id"] != q: j -= 1
    e = int(rows[j]["End_Timestamp"])
    side = [r for r in rows[max(0, i - 40):i] if r["Queue_Id"] != q and int(r["End_Timestamp"]) > e]
    print(f"gap before optimizer {(s - e) / 1e3:7.1f} us after {rows[j]['Kernel_Name'][:40]}; side kernels finishing inside: " +
          ", ".join(f"{r['Kernel_Name'].split('(')[0][-28:]}:{(int(r['End_Timestamp']) - max(e, int(r['Start_Timestamp']))) / 1e3:.0f}us" for r in side))
# and the start of the step: first conv after the optimizer
for i in idx[-6:-1]:
    e = int(rows[i]["End_Timestamp"])
    nxt = next(r for r in rows[i + 1:] if "conv_dma" in r["Kernel_Name"])
    k = [r for r in rows[i:i + 60] if int(r["Start_Timestamp"]) < int(nxt["Start_Timestamp"])]
    print(f"optimizer -> first conv of the next step: {(int(nxt['Start_Timestamp']) - e) / 1e3:7.1f} us, {len(k)} kernels between")

i = idx[-3]
e = int(rows[i]["Start_Timestamp"])
nxt = next(r for r in rows[i + 1:] if "conv_dma" in r["Kernel_Name"])
for r in rows[i:i + 40]:
    if int(r["Start_Timestamp"]) > int(nxt["End_Timestamp"]): break
    print(f"q{r['Queue_Id']} start {(int(r['Start_Timestamp']) - e) / 1e3:8.1f} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
