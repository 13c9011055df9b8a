import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import train_ops as T, ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, C, H, s) in ((4, 64, 97, 2), (4, 128, 49, 2), (4, 256, 25, 2), (4, 512, 13, 1)):
    x = torch.relu(torch.randn(N, C, H, H, generator=g)).requires_grad_()
    y = F.max_pool2d(x, 3, s, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    yy, idx = T.maxpool_idx(xn, 3, s, 1)
    dx = T.maxpool_idx_bwd(idx, dy.permute(0, 2, 3, 1).contiguous().to(dev), (H, H), 3, s, 1)
    ref = x.grad.permute(0, 2, 3, 1)
    print(N, C, H, s, "fwd equal", torch.equal(yy.cpu(), y.detach().permute(0, 2, 3, 1)), "bwd max diff", (dx.cpu() - ref).abs().max().item(),
          "sum hip", dx.double().sum().item(), "ref", ref.double().sum().item(), "dy sum", dy.double().sum().item())
