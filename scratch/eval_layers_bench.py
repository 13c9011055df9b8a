"""The heavy layers of the 25-episode eval step (M = 130 050) alone on the chip, per exact tile variant: us and TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
N = 50
for (cin, cout, k, d, res) in ((256, 1024, 1, 1, 1), (512, 1024, 1, 1, 0), (1024, 256, 1, 1, 0), (256, 256, 3, 2, 0), (128, 512, 1, 1, 1), (1024, 512, 1, 1, 0)):
    x = torch.randn(N, 51, 51, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), cin, cout, k, k, 1, d if k == 3 else 0, d, kpad, False, True)
    r = torch.randn(N, 51, 51, cout, device=dev) if res else None
    out = torch.empty(N, 51, 51, cout, device=dev)
    ref = ops.conv2d(x, prm, residual=r, tile=23).clone()
    fl = 2.0 * N * 51 * 51 * cout * k * k * cin
    row = []
    for tile in (23, 25, 24, 26, 27):
        if cout % ops.TILE_VARIANTS[tile][1]: continue
        same = torch.equal(ops.conv2d(x, prm, residual=r, out=out, tile=tile), ref)
        us = t(lambda: ops.conv2d(x, prm, residual=r, out=out, tile=tile))
        row.append(f"{tile}: {us:6.1f}us {fl / us / 1e6:5.1f}TF{'' if same else ' !'}")
    print(f"{cin:4d}->{cout:4d} k{k} d{d} res{res} | " + " | ".join(row), flush=True)
