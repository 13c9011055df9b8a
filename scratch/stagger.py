"""Does de-phasing the two co-resident blocks of a CU help the short-K conv layers?  flags bits 18..21 = sleep count."""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def t(fn, n=10):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, k, d, res) in [(256, 1024, 1, 1, True), (128, 512, 1, 1, True), (1024, 256, 1, 1, False), (256, 256, 3, 2, False)]:
    n, h, w = 50, 51, 51
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(wt)
    r = torch.randn(n, h, w, cout, device=dev) if res else None
    out = torch.empty(n, h, w, cout, device=dev)
    M = n * h * w
    fl = 2.0 * M * cout * k * k * cin
    for tile in (21, 24, 25, 26, 27):
        bm, bn = ops.TILE_VARIANTS[tile]
        if cout % bn: continue
        row = []
        ref = None
        for sl in (0, 1, 2, 4, 8):
            d_ = ops.ConvDesc(n, h, w, cin, cin, h, w, cout, cout, k, k, 1, d * (k // 2), d, cout if res else 0, kpad, 1 | (sl << 18), tile)
            def run():
                _lib.check(lib.pemp_conv2d_nhwc_f32(C.byref(d_), x.data_ptr(), packed.data_ptr(), out.data_ptr(), None, None,
                                                    r.data_ptr() if res else None, torch.cuda.current_stream().cuda_stream), "conv")
            us = t(run)
            if ref is None: ref = out.clone()
            else: assert torch.equal(out, ref)
            row.append(f"{fl/us/1e6:6.1f}")
        print(f"cin={cin} cout={cout} k={k} res={int(res)} tile {bm}x{bn}({tile}): sleep 0/1/2/4/8 -> " + " ".join(row), flush=True)
