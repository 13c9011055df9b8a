"""Every exact tile variant on the layer shapes of a one-episode step (M = 5202): us per launch, eager, best of 3 x 40."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def t(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for (cin, cout, k, d, res) in ((256, 256, 3, 2, 0), (1024, 256, 1, 1, 0), (256, 1024, 1, 1, 1), (512, 128, 1, 1, 0), (128, 128, 3, 1, 0), (128, 512, 1, 1, 1), (1280, 512, 1, 1, 0)):
    x = torch.randn(2, 51, 51, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, None, torch.zeros(cout, device=dev), cin, cout, k, k, 1, d if k == 3 else 0, d, kpad, False, True)
    r = torch.randn(2, 51, 51, cout, device=dev) if res else None
    ref = ops.conv2d(x, prm, residual=r, tile=23)
    row = []
    for tile in (23, 22, 25, 21, 24, 26, 27, 28, 29):
        if cout % ops.TILE_VARIANTS[tile][1]: continue
        same = torch.equal(ops.conv2d(x, prm, residual=r, tile=tile), ref)
        row.append(f"{tile}: {t(lambda: ops.conv2d(x, prm, residual=r, tile=tile)):6.1f}{'' if same else '!'}")
    fl = 2.0 * 5202 * cout * k * k * cin
    print(f"{cin:4d}->{cout:4d} k{k} d{d} res{res} ({fl / 157.3e6:5.1f} us at peak) | " + " | ".join(row))
