"""BN train kernels vs float64 torch on the host: python scratch/bn_check.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import train_ops as T
import torch.nn.functional as F
dev = torch.device("cuda:0")
for (M, C, relu, res) in ((338, 256, False, False), (338, 256, True, True), (5202, 1024, True, True), (20808, 256, True, False), (81608, 64, True, False)):
    g = torch.Generator().manual_seed(1)
    z = (torch.randn(M, C, generator=g) * 3 + 0.5)
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g)
    r = torch.randn(M, C, generator=g) if res else None
    dy = torch.randn(M, C, generator=g)
    zd = z.double().requires_grad_(); gd = gamma.double().requires_grad_(); bd = beta.double().requires_grad_()
    y = F.batch_norm(zd.t().reshape(1, C, M, 1), None, None, gd, bd, True, 0.1, 1e-5).reshape(C, M).t()
    if res: y = y + r.double()
    if relu: y = F.relu(y)
    y.backward(dy.double())
    zg = z.to(dev)
    mean, invstd = T.bn_stats(zg, 1e-5, 0.1, None, None)
    out = torch.empty(M, C, device=dev)
    T.bn_apply(zg, mean, invstd, gamma.to(dev), beta.to(dev), out, residual=r.to(dev) if res else None, relu=relu)
    dz = torch.empty(M, C, device=dev)
    dgamma, dbeta = T.bn_bwd(dy.to(dev), out, zg, mean, invstd, gamma.to(dev), dz, relu=relu)
    f = lambda a, b: ((a.cpu().double() - b).abs().max() / b.abs().max()).item()
    print(f"M={M} C={C} relu={relu} res={res}: y {f(out, y.detach()):.2e} dz {f(dz, zd.grad):.2e} dgamma {f(dgamma, gd.grad):.2e} dbeta {f(dbeta, bd.grad):.2e}")
