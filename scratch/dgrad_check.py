import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import train_ops as T, ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, Cin, Cout, H, d) in ((4, 64, 64, 97, 1), (4, 64, 128, 49, 1), (4, 128, 128, 49, 1), (4, 256, 256, 25, 1), (4, 512, 512, 13, 2)):
    x = torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64).requires_grad_()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) / (Cin * 9) ** 0.5).requires_grad_()
    y = F.conv2d(x, w, None, 1, d, d)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64) * (torch.rand(y.shape, generator=g, dtype=torch.float64) > 0.5)
    y.backward(dy)
    wk = w.detach().float().permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev)
    wd = T.dgrad_weight(wk, 3, 3)
    prm = ops.ConvParams(wd, None, None, Cout, Cin, 3, 3, 1, d * 2 - d, d, wd.shape[1], False, False)
    gd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    outs = {}
    for tile in (3, 13, 23, 25):
        dx = ops.conv2d(gd, prm, tile=tile)
        outs[tile] = dx
    dx = outs[23].cpu().double().permute(0, 3, 1, 2)
    ref = x.grad
    # torch fp32 on CPU for comparison
    x32 = x.detach().float().requires_grad_(); w32 = w.detach().float()
    F.conv2d(x32, w32, None, 1, d, d).backward(dy.float())
    rel = lambda a: ((a.double() - ref).norm() / ref.norm()).item()
    print(N, Cin, Cout, H, d, "L2 rel hip", f"{rel(dx):.2e}", "torch32", f"{rel(x32.grad):.2e}", "sum rel hip", f"{abs(dx.sum() - ref.sum()).item() / ref.abs().sum().item():.2e}",
          "torch32", f"{abs(x32.grad.double().sum() - ref.sum()).item() / ref.abs().sum().item():.2e}", "variants identical", all(torch.equal(outs[3], v) for v in outs.values()))
