import sys, torch, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net, sd = bench.build_model(dev)
pool = bench.episode_pool(dev, 1, B, 0, n_groups=1)
rec = []
orig = ops.conv2d
def timed(x, p, out=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, p, out=out, **kw); e1.record()
    n, ho, wo, co = y.shape
    rec.append((e0, e1, (n*ho*wo, co, p.kh*p.kw*p.cin, p.kh, p.stride, p.dil), 2.0*n*ho*wo*co*p.kh*p.kw*(3 if p.stem else p.cin)))
    return y
ops.conv2d = timed
ep = pool[0]
with torch.no_grad():
    for r in range(4):
        if r == 1: rec.clear()
        net.lowres(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, key, fl in rec:
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{'M':>7} {'N':>5} {'K':>5} k s d  cnt   ms/call   TF/s   share")
for key, (cnt, ms, fl) in agg.items():
    M, N, K, k, s, d = key
    print(f"{M:7d} {N:5d} {K:5d} {k} {s} {d:2d} {cnt//3:4d} {ms/cnt:9.4f} {fl/ms/1e9:7.1f} {ms/tot*100:6.1f}%")
print("total conv ms/step", tot/3)
