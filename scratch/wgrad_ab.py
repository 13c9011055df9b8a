"""A/B of two builds of the library on the weight-gradient shapes of the training step, same process order, same box:
python scratch/wgrad_ab.py  (PEMP_HIP_LIB selects the build; run once per build)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, train_ops as T
from pemp_amd.ops import ConvParams
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
n = 8
tot = 0.0
for (cin, cout, k, d, hh, cnt) in [(256, 256, 3, 2, 51, 10), (1024, 256, 1, 1, 51, 6), (256, 1024, 1, 1, 51, 6), (512, 1024, 1, 1, 51, 1),
                                   (1024, 512, 1, 1, 51, 1), (128, 128, 3, 1, 51, 4), (64, 64, 3, 1, 101, 3), (256, 64, 1, 1, 101, 2), (64, 256, 1, 1, 101, 4)]:
    x = torch.randn(n, hh, hh, cin, device=dev); g = torch.randn(n, hh, hh, cout, device=dev)
    dw = torch.empty(cout, k * k * cin, device=dev)
    p = ConvParams(None, None, None, cin, cout, k, k, 1, d * (k // 2), d, k * k * cin, False, False)
    fl = 2.0 * n * hh * hh * cout * k * k * cin
    ws = {}
    best = None
    for kind in ((2, 3) if cin % 128 == 0 and cout % 128 == 0 else (0,)):
        for nb in (512, 768):
            us = min(t(lambda: T.conv_wgrad(x, g, p, dw, ws_cache=ws, blocks=(kind, nb) if kind else nb)) for _ in range(2))
            if best is None or us < best[0]: best = (us, kind, nb)
    tot += best[0] * cnt
    print(f"wgrad cin={cin} cout={cout} k={k}: best {best[0]:7.1f} us (kind {best[1]}, {best[2]} blocks) = {fl / best[0] / 1e6:6.1f} TFLOP/s", flush=True)
print(f"weighted sum over the step's layers: {tot / 1e3:.3f} ms  [{os.environ.get('PEMP_HIP_LIB', 'default build')}]")
