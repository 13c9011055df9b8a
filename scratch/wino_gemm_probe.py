"""PROBE for DESIGN.md "what comes next" (1): the GEMM part of Winograd F(2x2,3x3) for one 3x3 256->256 layer of the eval step --
16 products [T x 256] x [256 x 256], T = 50 images x 4 sub-lattices x 13 x 13 tiles = 33800 -- on the existing 1x1 conv
kernels: 16 launches, and ONE launch over 16 T rows (what a grouped launch would cost at best).  Compare: the direct layer
takes ~1060 us."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
from pemp_amd.ops import ConvParams
dev = torch.device("cuda:0")
torch.manual_seed(0)
C = 256
w = torch.randn(C, C, device=dev) * 0.05            # [Cout, Cin] = KRSC of a 1x1 conv
p = ConvParams(w.contiguous(), None, None, C, C, 1, 1, 1, 0, 1, C, False, False)
def t(fn, n=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, shape in (("T=33800", (50, 26, 26, C)), ("16T", (800, 26, 26, C)), ("direct-M", (50, 51, 51, C))):
    x = torch.randn(shape, device=dev)
    out = torch.empty(shape, device=dev)
    us = t(lambda: ops.conv2d(x, p, out=out))
    M = shape[0] * shape[1] * shape[2]
    print(f"{name}: M={M} one launch {us:.1f} us = {2.0 * M * C * C / us / 1e6:.1f} TFLOP/s" + (f";  x16 = {16 * us:.0f} us" if name == "T=33800" else ""))
