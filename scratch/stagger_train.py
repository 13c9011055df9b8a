"""De-phasing sweep at the training shapes (8 images, 51 x 51): conv tiles x stagger, weight-gradient kinds x blocks x stagger."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, train_ops as T
from pemp_amd.ops import ConvParams
dev = torch.device("cuda:0")
ops.AUTOTUNE = False
def t(fn, n=8):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
STG = (0, 8, 16, 32, 48, 64, 96)
n, h, w = 8, 51, 51
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("conv", "both"):
    for (cin, cout, k, d, res) in [(256, 256, 3, 2, False), (1024, 256, 1, 1, False), (256, 1024, 1, 1, False), (256, 1024, 1, 1, True),
                                   (512, 1024, 1, 1, False), (1024, 512, 1, 1, False), (128, 128, 3, 1, False)]:
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
        packed, kpad = ops.pack_conv_weight(wt)
        p = ConvParams(packed, None, None, cin, cout, k, k, 1, d * (k // 2), d, kpad, False, False)
        r = torch.randn(n, h, w, cout, device=dev) if res else None
        out = torch.empty(n, h, w, cout, device=dev)
        fl = 2.0 * n * h * w * cout * k * k * cin
        for tile in (21, 31, 24, 34, 22, 32, 25, 35, 23):
            bm, bn = ops.TILE_VARIANTS[tile - 10 if tile > 30 else tile]
            if cout % bn: continue
            row = []
            for sl in STG:
                ops.STAGGER = sl
                row.append("%6.1f" % (fl / t(lambda: ops.conv2d(x, p, out=out, residual=r, tile=tile, splitk=True)) / 1e6))
            ops.STAGGER = 0
            print(f"conv cin={cin} cout={cout} k={k} res={int(res)} tile {tile} ({bm}x{bn}): " + " ".join(row), flush=True)
if which in ("wgrad", "both"):
    for (cin, cout, k, d) in [(256, 256, 3, 2), (1024, 256, 1, 1), (256, 1024, 1, 1), (512, 1024, 1, 1), (128, 128, 3, 1)]:
        x = torch.randn(n, h, w, cin, device=dev)
        g = torch.randn(n, h, w, cout, device=dev)
        dw = torch.empty(cout, k * k * cin, device=dev)
        p = ConvParams(None, None, None, cin, cout, k, k, 1, d * (k // 2), d, k * k * cin, False, False)
        fl = 2.0 * n * h * w * cout * k * k * cin
        ws = {}
        for kind in (2, 3):
            for nb in (256, 384, 512, 768):
                row = []
                for sl in STG:
                    ops.WGRAD_STAGGER = sl
                    row.append("%6.1f" % (fl / t(lambda: T.conv_wgrad(x, g, p, dw, ws_cache=ws, blocks=(kind, nb))) / 1e6))
                ops.WGRAD_STAGGER = 0
                print(f"wgrad cin={cin} cout={cout} k={k} kind {kind} blocks {nb}: " + " ".join(row), flush=True)
