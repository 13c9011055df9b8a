"""The layer-1 / stem-side convs of the 25-episode eval step (50 images, 101 x 101: M = 510 050 rows) alone on the chip: us per launch
per exact tile variant, and what that is in HBM bytes per second (algorithmic bytes: input + output + shortcut + weights)."""
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
for (cin, cout, k, res) in ((64, 64, 1, 0), (64, 64, 3, 0), (64, 256, 1, 0), (64, 256, 1, 1), (256, 64, 1, 0)):
    x = torch.randn(N, 101, 101, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), cin, cout, k, k, 1, 1 if k == 3 else 0, 1, kpad, False, True)
    r = torch.randn(N, 101, 101, cout, device=dev) if res else None
    out = torch.empty(N, 101, 101, cout, device=dev)
    M = N * 101 * 101
    nbytes = 4.0 * M * (cin + cout * (2 if res else 1))
    fl = 2.0 * M * cout * k * k * cin
    row = []
    for tile in (23, 22, 25, 21, 24, 26, 27, 13, 12, 15):
        if cout % ops.TILE_VARIANTS[tile][1]: continue
        us = t(lambda: ops.conv2d(x, prm, residual=r, out=out, tile=tile))
        row.append(f"{tile}: {us:6.1f}us {nbytes / us / 1e6:4.2f}TB/s")
    print(f"{cin:4d}->{cout:4d} k{k} res{res} ({nbytes / 1e6:6.0f} MB, {fl / 1e9:5.1f} GFLOP = {fl / 157.3e6:5.1f} us at MFMA peak, {nbytes / 5.5e6:5.1f} us at 5.5 TB/s) | " + " | ".join(row), flush=True)
