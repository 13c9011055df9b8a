"""A/B of the two-stream encoder on one-episode steps: episodes/s at one in flight, halves on / off; outputs bit-identical."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pemp_amd import ops
dev = torch.device("cuda:0")
net, sd = bench.build_model(dev)
pool = bench.episode_pool(dev, 1, 1, 0, n_groups=5)
def run(n=200):
    outs = []
    with torch.no_grad():
        for i in range(10):
            ep = pool[i % len(pool)]
            pred, _ = net.lowres_graphed(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
            outs.append(pred.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            ep = pool[i % len(pool)]
            pred, _ = net.lowres_graphed(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return n / dt, outs[:5]
res = {}
for halves in (False, True, False, True):
    ops.ENCODE_HALVES = halves
    eps, outs = run()
    res.setdefault(halves, []).append(eps)
    if halves: oh = outs
    else: ob = outs
    print(f"halves={halves}: {eps:.1f} episodes/s ({1e3 / eps:.3f} ms)")
print("bit-identical:", all(torch.equal(a, b) for a, b in zip(oh, ob)))
