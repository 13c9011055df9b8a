"""Isolated launches of the 25-episode eval step's 256 -> 1024 + shortcut layer (M = 130 050) and, for contrast, the 3 x 3 256 -> 256
layer, on the tiles the autotuner weighs (24: 128 x 128 8-wave, 27: 256 x 256), for PMC passes (KPROBE=scratch/kprobe_eval.py)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
from pemp_amd.ops import ConvParams
dev = torch.device("cuda:0")
ops.AUTOTUNE = False
REPS = int(os.environ.get("REPS", "6"))
def conv_case(n, cin, cout, k, d, res, tile):
    x = torch.randn(n, 51, 51, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    packed, kpad = ops.pack_conv_weight(w)
    p = ConvParams(packed, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), cin, cout, k, k, 1, d * (k // 2), d, kpad, False, True)
    r = torch.randn(n, 51, 51, cout, device=dev) if res else None
    out = torch.empty(n, 51, 51, cout, device=dev)
    for _ in range(REPS):
        ops.conv2d(x, p, out=out, residual=r, tile=tile)
conv_case(50, 256, 1024, 1, 1, True, 24)
conv_case(50, 256, 1024, 1, 1, False, 24)
conv_case(50, 256, 1024, 1, 1, True, 27)
conv_case(50, 1024, 256, 1, 1, False, 27)
conv_case(50, 256, 256, 3, 2, False, 27)
torch.cuda.synchronize()
