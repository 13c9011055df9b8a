"""Seeded fuzzing of the conv engine over random geometries (ragged M, odd spatial sizes, stride / dilation / padding
combinations, channel counts that are multiples of 32 but not of the tile sizes' K chunks, every tile variant):
 * forward (+ affine + residual + ReLU) vs torch CPU conv2d,
 * every applicable kernel variant BIT-IDENTICAL to the first (the autotuner may pick any of them),
 * weight gradient and input gradient vs torch autograd."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TILE_N = {3: 64, 2: 64, 1: 128, 13: 64, 12: 64, 11: 128, 15: 64, 14: 128, 16: 128, 17: 256,
          23: 64, 22: 64, 21: 128, 25: 64, 24: 128, 26: 128, 27: 256, 28: 64, 29: 64}     # 2x: conv_dma2.hip (buffer-addressed LDS-DMA)


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice((1, 1, 3, 3, 5))
        d = rng.choice((1, 1, 2, 3, 6))
        s = rng.choice((1, 1, 1, 2))
        p = rng.choice((0, d * (k // 2), d * (k // 2) + 1, 1))
        H, W = rng.randint(3, 37), rng.randint(3, 37)
        if (H + 2 * p - d * (k - 1) - 1) // s + 1 < 1 or (W + 2 * p - d * (k - 1) - 1) // s + 1 < 1 or H + 2 * p < d * (k - 1) + 1 or W + 2 * p < d * (k - 1) + 1:
            continue
        out.append((rng.randint(1, 5), H, W, rng.choice((32, 64, 96, 160, 256)), rng.choice((64, 128, 192, 256, 320, 512)), k, s, p, d))
    return out


def _rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("case", _cases(24, 2026), ids=lambda c: "x".join(str(v) for v in c))
def test_conv_forward_fuzz(hip_lib, dev, case):
    from pemp_amd import ops
    N, H, W, Cin, Cout, k, s, p, d = case
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, k, k, seed=2) / (Cin * k * k) ** 0.5
    scale, shift = torch.rand(Cout, generator=torch.Generator().manual_seed(3)) + 0.5, _rand(Cout, seed=4)
    ref = F.conv2d(x, w, None, s, p, d)
    res = _rand(*ref.shape, seed=5)
    ref = F.relu(ref * scale[None, :, None, None] + shift[None, :, None, None] + res)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, scale.to(dev), shift.to(dev), Cin, Cout, k, k, s, p, d, kpad, False, True)
    xd, rd = x.permute(0, 2, 3, 1).contiguous().to(dev), res.permute(0, 2, 3, 1).contiguous().to(dev)
    first = None
    for tile, need in TILE_N.items():
        if Cout % need:
            continue
        y = ops.conv2d(xd, prm, residual=rd, tile=tile)
        if first is None:
            first = y.clone()
            got = y.permute(0, 3, 1, 2).cpu()
            tol = 3e-5 * max(1.0, (Cin * k * k / 64) ** 0.5)
            assert ((got - ref).abs() / (1 + ref.abs())).max().item() < tol, case
        else:
            assert torch.equal(y, first), (case, tile)


@pytest.mark.parametrize("case", _cases(12, 77), ids=lambda c: "x".join(str(v) for v in c))
def test_conv_backward_fuzz(hip_lib, dev, case):
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, s, p, d = case
    Cin = (Cin + 63) // 64 * 64                 # the weight-gradient kernel (and dgrad's Cout) work on 64-channel groups
    if s == 2 and k != 1:
        s = 1                                   # the path's strided convs are 1x1 (backbones.py:47,110)
    if d * (k - 1) - p < 0:
        p = d * (k // 2)                        # dgrad-as-forward needs pad' = d(k-1) - p >= 0 (true for every layer of the path)
    x = _rand(N, Cin, H, W, seed=1).requires_grad_()
    w = (_rand(Cout, Cin, k, k, seed=2) / (Cin * k * k) ** 0.5).requires_grad_()
    y = F.conv2d(x, w, None, s, p, d)
    g = _rand(*y.shape, seed=3)
    y.backward(g)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    prm = ops.ConvParams(None, None, None, Cin, Cout, k, k, s, p, d, k * k * Cin, False, False)
    dw = torch.empty((Cout, k * k * Cin), device=dev)
    T.conv_wgrad(nhwc(x.detach()), nhwc(g), prm, dw)
    got = dw.cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    terms = N * y.shape[2] * y.shape[3]
    assert ((got - w.grad).abs() / (1 + w.grad.abs())).max().item() < 4e-5 * max(1.0, (terms / 64) ** 0.5), case
    wk, _ = ops.pack_conv_weight(w.detach().to(dev))
    pd = ops.ConvParams(T.dgrad_weight(wk, k, k), None, None, Cout, Cin, k, k, 1, d * (k - 1) - p, d, k * k * Cout, False, False)
    dx = ops.conv2d(nhwc(g), pd)
    if s != 1:
        dx = T.scatter_strided(dx, (H, W), s)
    ref = x.grad
    if s == 1 and tuple(dx.shape[1:3]) != (H, W):
        pytest.skip("geometry whose input gradient is not a same-size conv")
    assert ((dx.permute(0, 3, 1, 2).cpu() - ref).abs() / (1 + ref.abs())).max().item() < 4e-5 * max(1.0, (Cout * k * k / 64) ** 0.5), case


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, k, s, p, d      (Wo >= 32: the shapes the buffer-addressed weight-gradient kernels take)
    (2, 51, 51, 256, 256, 3, 1, 2, 2), (3, 51, 51, 128, 512, 1, 1, 0, 1), (2, 101, 101, 64, 64, 3, 1, 1, 1),
    (2, 101, 101, 256, 128, 1, 2, 0, 1), (1, 40, 33, 128, 256, 3, 1, 6, 6), (2, 37, 45, 64, 192, 3, 1, 1, 1),
    (1, 32, 32, 1024, 256, 1, 1, 0, 1), (2, 33, 64, 128, 128, 5, 1, 2, 1)], ids=lambda c: "x".join(str(v) for v in c))
def test_wgrad_generations_bit_identical_and_match_autograd(hip_lib, dev, case):
    """conv_wgrad2_kernel (buffer-addressed LDS-DMA, incremental pixel walk, barrier inside the MFMA stream) against the
    first-generation kernels bit for bit (same split, same reduction order) and against torch autograd; plain and
    accumulating, ragged M (not a multiple of the 32-pixel step)."""
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, s, p, d = case
    x = _rand(N, Cin, H, W, seed=1).requires_grad_(False)
    w = (_rand(Cout, Cin, k, k, seed=2) / (Cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p, d)
    g = _rand(*y.shape, seed=3)
    (gw,) = torch.autograd.grad(y, w, g)
    ref = gw.permute(0, 2, 3, 1).reshape(Cout, -1)
    prm = ops.ConvParams(None, None, None, Cin, Cout, k, k, s, p, d, k * k * Cin, False, False)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    gd = g.permute(0, 2, 3, 1).contiguous().to(dev)
    outs = []
    for variant in (1, 0):
        dw = torch.full((Cout, k * k * Cin), float("nan"), device=dev)
        T.conv_wgrad(xd, gd, prm, dw, variant=variant, blocks=0)          # the library's split for both: same reduction order
        outs.append(dw.clone())
        T.conv_wgrad(xd, gd, prm, dw, accumulate=True, variant=variant, blocks=0)
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    scale = ref.abs().max().item()
    assert (outs[2].cpu() - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (N * y.shape[2] * y.shape[3] / 4096) ** 0.5)
    assert (outs[3].cpu() - 2 * ref).abs().max().item() <= 4e-5 * scale * max(1.0, (N * y.shape[2] * y.shape[3] / 4096) ** 0.5)
    # every (tile kind, block count) the autotuner may pick gives the same gradient up to the rounding of the regrouped sum
    kinds = (2, 3) if Cin % 128 == 0 and Cout % 128 == 0 else (0,)
    for kind in kinds:
        for nb in T.WGRAD_BLOCK_CHOICES:
            dw = torch.full((Cout, k * k * Cin), float("nan"), device=dev)
            T.conv_wgrad(xd, gd, prm, dw, blocks=(kind, nb) if kind else nb)
            assert (dw.cpu() - ref).abs().max().item() <= 2e-5 * scale * max(1.0, (N * y.shape[2] * y.shape[3] / 4096) ** 0.5), (kind, nb)
