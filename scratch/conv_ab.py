"""A/B of the conv kernel families on the layer shapes of the stage-1 network: python scratch/conv_ab.py B
For each distinct conv geometry of one eval step (B episodes) time every tile variant of conv_dma.hip (1x) and
conv_dma2.hip (2x); print the best of each family, the ratio, and whether all outputs are bit-identical."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net, sd = bench.build_model(dev)
pool = bench.episode_pool(dev, 1, B, 0, n_groups=1)
seen = {}
orig = ops.conv2d
def spy(x, p, out=None, **kw):
    y = orig(x, p, out=out, **kw)
    if not p.stem and kw.get("pad_value") is None:
        key = (tuple(x.shape), p.cout, p.kh, p.stride, p.pad, p.dil, kw.get("residual") is not None)
        if key not in seen:
            seen[key] = (x.clone(), p, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
    return y
ops.conv2d = spy
import pemp_amd.engine as eng
eng.ops.conv2d = spy
ep = pool[0]
with torch.no_grad():
    net.lowres(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
ops.conv2d = orig
def t(fn, n=5):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot1 = tot2 = 0.0
print(f"{'M':>7} {'Cin':>5} {'N':>5} k s  d res | best1x us (tile) | best2x us (tile) | ratio | TF2x | identical")
for key, (x, p, kw) in seen.items():
    n, h, w, cin = x.shape
    res = {}
    ref = None; same = True
    for tile, (bm, bn) in ops.TILE_VARIANTS.items():
        if p.cout % bn or tile < 10: continue
        y = orig(x, p, tile=tile, **kw)
        if ref is None: ref = y.clone()
        else: same = same and torch.equal(y, ref)
        res[tile] = t(lambda: orig(x, p, tile=tile, **kw))
    b1 = min((v, k) for k, v in res.items() if k < 20); b2 = min((v, k) for k, v in res.items() if k >= 20)
    ho = ops.conv_out_size(h, p.kh, p.stride, p.pad, p.dil); M = n * ho * ops.conv_out_size(w, p.kw, p.stride, p.pad, p.dil)
    fl = 2.0 * M * p.cout * p.kh * p.kw * cin
    print(f"{M:7d} {cin:5d} {p.cout:5d} {p.kh} {p.stride} {p.dil:2d} {int(key[-1])}   | {b1[0]:8.1f} ({b1[1]}) | {b2[0]:8.1f} ({b2[1]}) | {b1[0]/b2[0]:5.3f} | {fl/b2[0]/1e6:5.1f} | {same}")
