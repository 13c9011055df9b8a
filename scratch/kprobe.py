"""Isolated launches of the training-shape GEMM kernels for PMC passes: python scratch/kprobe.py
wgrad 3x3 256->256 (128 x 128 tiles, 512 blocks), its forward conv with split-K remainder (tile 31 / 34), the 1x1 1024->256
pair, and -- for reference -- the eval-shape 3x3 conv on the 256 x 256 tile."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, train_ops as T
from pemp_amd.ops import ConvParams
dev = torch.device("cuda:0")
ops.AUTOTUNE = False
REPS = int(os.environ.get("REPS", "10"))
def conv_case(n, cin, cout, k, d, tile):
    x = torch.randn(n, 51, 51, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    p = ConvParams(packed, None, None, cin, cout, k, k, 1, d * (k // 2), d, kpad, False, False)
    out = torch.empty(n, 51, 51, cout, device=dev)
    for _ in range(REPS):
        ops.conv2d(x, p, out=out, tile=tile, splitk=True)
def wgrad_case(n, cin, cout, k, d, kind, nb):
    x = torch.randn(n, 51, 51, cin, device=dev)
    g = torch.randn(n, 51, 51, cout, device=dev)
    dw = torch.empty(cout, k * k * cin, device=dev)
    p = ConvParams(None, None, None, cin, cout, k, k, 1, d * (k // 2), d, k * k * cin, False, False)
    ws = {}
    for _ in range(REPS):
        T.conv_wgrad(x, g, p, dw, ws_cache=ws, blocks=(kind, nb))
wgrad_case(8, 256, 256, 3, 2, 2, 512)
wgrad_case(8, 1024, 256, 1, 1, 2, 512)
conv_case(8, 256, 256, 3, 2, 31)
conv_case(8, 256, 256, 3, 2, 34)
conv_case(8, 1024, 256, 1, 1, 31)
conv_case(50, 256, 256, 3, 2, 27)
torch.cuda.synchronize()
