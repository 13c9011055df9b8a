"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) of bench.py into
profiles/<name>.json: HBM bytes per conv launch, gfx950 FETCH_SIZE correction applied (x2, see
/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys
fdir, wdir, out, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def avg(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and ("conv_dma_kernel" in r["Kernel_Name"] or "conv_igemm_kernel" in r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
            n += 1
    return tot / n, n


fk, n = avg(fdir, "FETCH_SIZE")
wk, _ = avg(wdir, "WRITE_SIZE")
rec = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 5 --warmup 2 "
                  "--cpu-episodes 0 --no-roofline --no-e2e (%d episodes/step)" % batch,
       "episodes_per_step": batch,
       "kernels": "conv_dma_kernel* + conv_igemm_kernel* (%d launches; replay tile picks with PEMP_TILE_CACHE to keep autotune launches out)" % n,
       "FETCH_SIZE_KB_avg_per_launch": round(fk, 2), "WRITE_SIZE_KB_avg_per_launch": round(wk, 2),
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
       "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(rec, open(out, "w"), indent=1)
print(rec)
