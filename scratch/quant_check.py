"""How much of the training convs' distance to peak is tile quantisation?  Times each conv_dma2 tile on the layer-3 shapes
at M = 20808 (8 images of 51 x 51) and at an M that fills 256 CUs evenly."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, k, d) in [(256, 256, 3, 2), (1024, 256, 1, 1), (256, 1024, 1, 1), (512, 128, 1, 1), (128, 128, 3, 1)]:
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, None, None, cin, cout, k, k, 1, d * (k // 2), d, kpad, False, False)
    for (n, h, ww) in [(8, 51, 51), (8, 64, 64), (8, 64, 32), (8, 128, 32)]:
        x = torch.randn(n, h, ww, cin, device=dev)
        M = n * h * ww
        fl = 2.0 * M * cout * k * k * cin
        row = []
        for tile in range(21, 28):
            bm, bn = ops.TILE_VARIANTS[tile]
            if cout % bn: continue
            us = t(lambda: ops.conv2d(x, prm, tile=tile))
            blocks = -(-M // bm) * (cout // bn)
            row.append(f"{bm}x{bn}:{fl/us/1e6:5.1f}TF({blocks/256:.2f})")
        print(f"cin={cin} cout={cout} k={k} M={M}: " + "  ".join(row), flush=True)
