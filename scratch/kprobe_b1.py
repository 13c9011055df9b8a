"""Isolated launches of the one-episode conv shapes (M = 5202) on the exact tile variants 23 / 25 / 28, for PMC passes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
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
        ops.conv2d(x, p, out=out, tile=tile)
for tile in (23, 25, 28):
    conv_case(2, 256, 256, 3, 2, tile)
for tile in (23, 28):
    conv_case(2, 1024, 256, 1, 1, tile)
torch.cuda.synchronize()
