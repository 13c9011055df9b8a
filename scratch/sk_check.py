"""Per-tile timing of the training convs (stats epilogue) at the layer-3 shapes: plain (21..27) against split-K (31..37)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
import ctypes as C
from pemp_amd import _lib
DBG = int(os.environ.get('SKDBG','0'))
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, k, d) in [(256, 256, 3, 2), (1024, 256, 1, 1)]:
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, None, None, cin, cout, k, k, 1, d * (k // 2), d, kpad, False, False)
    x = torch.randn(8, 51, 51, cin, device=dev)
    M = 8 * 51 * 51
    fl = 2.0 * M * cout * k * k * cin
    row = []
    for tile in list(range(21, 28)) + list(ops.SPLITK_TILES):
        bm, bn = ops.TILE_VARIANTS[tile - 10 if tile > 30 else tile]
        if cout % bn: continue
        us = t(lambda: ops.conv2d_stats(x, prm, tile=tile))
        row.append(f"{tile}:{fl/us/1e6:5.1f}")
    print(f"cin={cin} cout={cout} k={k}: " + "  ".join(row), flush=True)
