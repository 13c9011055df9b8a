"""Streaming rate of the prototype-head kernels (cosine map, MPM assign + pool) at the bench shapes.

    python scratch/head_bench.py [B]            # PEMP_HEAD_VALU=1 selects the wave-per-pixel variants

Prints per-call time and algorithmic GB/s (feature-map bytes / time) for each op.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pemp_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
h = w = 51
c, p = 512, 3
H = W = 401
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
feat = torch.randn(B, h, w, c, generator=g).relu_().to(dev)
qry = torch.randn(B, h, w, c, generator=g).relu_().to(dev)
mask = (torch.rand(B, 1, H, W, generator=g) > 0.6).float()
mask = torch.cat([mask, 1 - mask], 1).contiguous().to(dev)
ctr = torch.randn(c, 2 * p, generator=g).to(dev)
cache = {}
protos = ops.mpm_protos(feat, mask, ctr, B, 1, p, ws_cache=cache)
fbytes = feat.numel() * 4


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


pred = torch.empty(B, 2, h, w, device=dev)
resp = torch.empty(B, h, w, dtype=torch.uint8, device=dev)
t = timeit(lambda: ops.cosine_proto_max(qry, protos, 20.0, want_resp=True, pred=pred, resp=resp))
print(f"cosine      B={B}: {t:8.1f} us  {fbytes / t / 1e3:7.1f} GB/s")
out = torch.empty(B, 2 * p, c, device=dev)
t = timeit(lambda: ops.mpm_protos(feat, mask, ctr, B, 1, p, ws_cache=cache, out=out))
print(f"mpm_protos  B={B}: {t:8.1f} us  {2 * fbytes / t / 1e3:7.1f} GB/s (two passes over the features)")
out2 = torch.empty(B, 2, c, device=dev)
t = timeit(lambda: ops.masked_avg_pool(feat, mask, B, 1, False, ws_cache=cache, out=out2))
print(f"map         B={B}: {t:8.1f} us  {fbytes / t / 1e3:7.1f} GB/s")
