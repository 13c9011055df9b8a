import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
N, H, ci, co, k, d, tile = [int(v) for v in sys.argv[1:8]]
x = torch.randn(N, H, H, ci, device=dev)
w = torch.randn(co, ci, k, k, device=dev) * 0.05
pk, kpad = ops.pack_conv_weight(w)
p = ops.ConvParams(pk, None, None, ci, co, k, k, 1, d * (k // 2), d, kpad, False, True)
out = ops.conv2d(x, p, tile=tile)
for _ in range(5):
    ops.conv2d(x, p, out=out, tile=tile)
torch.cuda.synchronize()
