"""kernel_trace.csv -> kernels of the last complete step (between two pack_input launches of the FIRST queue seen), with queue id,
start, end: shows whether two streams really overlap.  python scratch/step_sequence2.py trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "eval_tail_final" in r["Kernel_Name"] or "cosine_mfma" in r["Kernel_Name"]]
# a step = from the kernel after the previous cosine to this cosine
cos = [i for i, r in enumerate(rows) if "cosine_mfma" in r["Kernel_Name"]]
a, b = cos[-3] + 1, cos[-2] + 1
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"])
qs = {}
for r in seq:
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    qs.setdefault(q, len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void pemp::", "").replace("pemp::", "")[:60]
    print(f"q{qs[q]} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f} us)  grid {r.get('Grid_Size', ''):>8}  {name}")
print("wall", (int(seq[-1]["End_Timestamp"]) - t0) / 1e3, "us;", len(seq), "kernels; queues", len(qs))
