"""Which tile variant the autotuner picked for every conv / weight-gradient shape of the stage-1 training step, and why the
64 x 64 kernels are still in it (verdict r3, item 3b).  Reads the pick file a training run wrote (PEMP_TILE_CACHE):
    PEMP_TILE_CACHE=$PWD/gpurun_out/r04/train_tiles.json python3 bench.py --mode train --steps 3 --warmup 2 --cpu-episodes 0 --no-roofline
    python3 scratch/train_picks.py gpurun_out/r04/train_tiles.json"""
import json
import sys

SHAPE = {1: (128, 128, 4), 2: (128, 64, 4), 3: (64, 64, 4), 4: (128, 128, 8), 5: (128, 64, 8), 6: (256, 128, 8), 7: (256, 256, 8)}
KIND = {0: "conv", 2: "conv + BN statistics", 3: "input gradient + BN backward sums", 4: "conv (split-K allowed)"}
rows = []
for k, v in json.load(open(sys.argv[1])).items():
    key = json.loads(k)
    if key[0] == "wgrad":
        rows.append(("wgrad", key[1:], v))
        continue
    if key[0] == -7:
        continue
    cin, cout, kh, kw, stride, pad, dil, kind, n, h, w, res, padv = key
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    m = n * ho * wo
    t = int(v)
    sk = t > 30
    bm, bn, waves = SHAPE[(t - 30) if sk else (t - 20) if t > 20 else (t - 10) if t > 10 else t]
    tiles = -(-m // bm) * (cout // bn)
    rows.append(("conv", (m, cin, cout, kh, dil, str(KIND.get(kind, kind)), res), (t, bm, bn, waves, sk, tiles, round(tiles / 256, 2))))
print(f"{'M':>7} {'Cin':>5} {'Cout':>5} k d  {'kind':36s} res | tile  BMxBN waves splitK  tiles rounds")
for kind, key, v in sorted([r for r in rows if r[0] == 'conv'], key=lambda r: (-r[1][0], r[1][1], r[1][2], r[1][3])):
    m, cin, cout, kh, dil, kd, res = key
    t, bm, bn, waves, sk, tiles, rounds = v
    print(f"{m:7d} {cin:5d} {cout:5d} {kh} {dil:2d}  {kd:36s} {res}   | {t:4d} {bm:4d}x{bn:<4d} {waves}     {str(sk):5s} {tiles:6d} {rounds:6.2f}")
print("\nweight gradients (key -> (tile kind, blocks)):")
for kind, key, v in rows:
    if kind == "wgrad":
        print("  ", key, "->", v)
