"""Parity of the training entry points of the C ABI against torch autograd on the CPU.

Tolerances: contractions over K (dgrad) or M (wgrad) terms are compared with
|d| <= 3e-5 * sqrt(terms/64) * (1 + |ref|); streaming kernels 1e-5 relative."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _close(got, ref, tol):
    return ((got - ref).abs() / (1 + ref.abs())).max().item() < tol


WG_CASES = [  # N, H, W, Cin, Cout, k, s, p, d
    (2, 13, 13, 64, 64, 1, 1, 0, 1),
    (2, 25, 25, 256, 128, 1, 2, 0, 1),
    (2, 13, 13, 128, 128, 3, 1, 2, 2),
    (1, 21, 17, 64, 256, 3, 1, 6, 6),
    (3, 9, 11, 128, 64, 3, 1, 1, 1),
    (8, 1, 1, 256, 256, 1, 1, 0, 1),
]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad_and_dgrad(hip_lib, dev, case):
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, s, p, d = case
    x = _rand(N, Cin, H, W, seed=1).requires_grad_()
    w = (_rand(Cout, Cin, k, k, seed=2) * (1.0 / (Cin * k * k) ** 0.5)).requires_grad_()
    y = F.conv2d(x, w, None, s, p, d)
    g = _rand(*y.shape, seed=3)
    y.backward(g)
    prm = ops.ConvParams(None, None, None, Cin, Cout, k, k, s, p, d, k * k * Cin, False, False)
    dw = torch.full((Cout, k * k * Cin), 7.0, device=dev)
    T.conv_wgrad(_nhwc(x.detach()).to(dev), _nhwc(g).to(dev), prm, dw)
    got = dw.cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    terms = N * y.shape[2] * y.shape[3]
    assert _close(got, w.grad, 3e-5 * max(1.0, (terms / 64) ** 0.5)), (got - w.grad).abs().max()
    # accumulate=True adds on top
    T.conv_wgrad(_nhwc(x.detach()).to(dev), _nhwc(g).to(dev), prm, dw, accumulate=True)
    assert _close(dw.cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2), 2 * w.grad, 6e-5 * max(1.0, (terms / 64) ** 0.5))
    # dgrad through the forward kernel with the flipped/transposed weight (stride 1) or + scatter (stride 2)
    wk, _ = ops.pack_conv_weight(w.detach().to(dev))
    wd = T.dgrad_weight(wk, k, k)
    pd = ops.ConvParams(wd, None, None, Cout, Cin, k, k, 1, d * (k - 1) - p, d, k * k * Cout, False, False)
    gd = _nhwc(g).to(dev)
    if s == 1:
        dx = ops.conv2d(gd, pd)
    else:
        dx = T.scatter_strided(ops.conv2d(gd, pd), (H, W), s)
    assert _close(_nchw(dx.cpu()), x.grad, 3e-5 * max(1.0, (Cout * k * k / 64) ** 0.5))


def test_conv_wgrad_stem(hip_lib, dev):
    from pemp_amd import ops, train_ops as T
    N, H, Cout, k = 2, 37, 64, 7
    x = _rand(N, 3, H, H, seed=1)
    w = (_rand(Cout, 3, k, k, seed=2) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, 2, 3)
    g = _rand(*y.shape, seed=3)
    y.backward(g)
    x4 = torch.zeros(N, H, H, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    prm = ops.ConvParams(None, None, None, 4, Cout, k, k, 2, 3, 1, 256, True, False)
    dw = torch.empty((Cout, 256), device=dev)
    T.conv_wgrad(x4.to(dev), _nhwc(g).to(dev), prm, dw)
    got = dw.cpu()[:, :196].view(Cout, k, k, 4)[..., :3].permute(0, 3, 1, 2)
    assert _close(got, w.grad, 2e-4)
    assert dw.cpu()[:, 196:].abs().max() == 0 and dw.cpu()[:, :196].view(Cout, 49, 4)[..., 3].abs().max() == 0


@pytest.mark.parametrize("M,C,relu,res", [(700, 64, True, False), (2601, 256, True, True), (300, 128, False, False), (5, 256, False, False)])
def test_bn_train_forward_backward(hip_lib, dev, M, C, relu, res):
    from pemp_amd import train_ops as T
    z = (_rand(M, C, seed=1) * 3 + 0.5).requires_grad_()
    gamma = _rand(C, seed=2, lo=0.5, hi=1.5).requires_grad_()
    beta = _rand(C, seed=3).requires_grad_()
    r = _rand(M, C, seed=4).requires_grad_() if res else None
    rm, rv = _rand(C, seed=5), _rand(C, seed=6, lo=0.5, hi=1.5)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    zz = z.t().reshape(1, C, M, 1)
    y = F.batch_norm(zz, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5).reshape(C, M).t()
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    dy = _rand(M, C, seed=7)
    y.backward(dy)
    zd = z.detach().to(dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    mean, invstd = T.bn_stats(zd, 1e-5, 0.1, rmd, rvd)
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    out = torch.empty(M, C, device=dev)
    T.bn_apply(zd, mean, invstd, gamma.detach().to(dev), beta.detach().to(dev), out, residual=r.detach().to(dev) if res else None, relu=relu)
    assert torch.allclose(out.cpu(), y.detach(), rtol=1e-5, atol=2e-5)
    dz = torch.empty(M, C, device=dev)
    gout = torch.empty(M, C, device=dev) if res else None
    dgamma, dbeta = T.bn_bwd(dy.to(dev), out, zd, mean, invstd, gamma.detach().to(dev), dz, gout=gout, relu=relu)
    assert torch.allclose(dz.cpu(), z.grad, rtol=1e-4, atol=2e-5), (dz.cpu() - z.grad).abs().max()
    assert torch.allclose(dgamma.cpu(), gamma.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dbeta.cpu(), beta.grad, rtol=1e-4, atol=1e-4)
    if res:
        assert torch.allclose(gout.cpu(), r.grad, rtol=1e-6, atol=1e-7)


STATS_CASES = [  # N, H, W, Cin, Cout, k, s, p, d
    (2, 51, 51, 64, 64, 1, 1, 0, 1),        # M = 5202: the last 32-row group is ragged (18 rows)
    (2, 51, 51, 64, 256, 3, 1, 1, 1),
    (3, 26, 26, 256, 512, 1, 2, 0, 1),
    (2, 26, 26, 128, 128, 3, 1, 2, 2),
    (1, 5, 3, 64, 64, 1, 1, 0, 1),          # M = 15 < one row group
    (8, 51, 51, 256, 256, 3, 1, 2, 2),      # a layer-3 conv of the training step: 326 tiles of 128 x 128 -> 70 split 3 ways
    (8, 51, 51, 256, 1024, 1, 1, 0, 1),     # 1304 tiles of 128 x 128 -> 24 split into 2 (8 K steps)
]


@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_with_batch_statistics_in_the_epilogue(hip_lib, dev, case):
    """pemp_conv2d_stats_nhwc_f32: z is bit-identical to the plain conv on every tile; the partial sums are the column
    sums of the stored z over each row tile (fp32 sums: 2e-6 relative to sum |z|), reproducible; mean / invstd /
    running statistics from them equal pemp_bn_stats_f32 of z to fp32 rounding, and pemp_bn_fwd_partials_f32 (statistics
    + normalisation + sign mask in one call) equals the two separate calls bit for bit."""
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, s, p, d = case
    x = _nhwc(_rand(N, Cin, H, W, seed=1)).to(dev)
    w = _rand(Cout, Cin, k, k, seed=2, lo=-0.1, hi=0.1)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, Cin, Cout, k, k, s, p, d, kpad, False, False)
    assert ops.stats_supported(x, prm)
    z_ref = ops.conv2d(x, prm, tile=13)
    M = z_ref.numel() // Cout

    def tile_sums(t, bm):          # [row tiles of bm rows, Cout] column sums (float64) of an [M, Cout] tensor
        pad = (-M) % bm
        return torch.cat([t, torch.zeros(pad, Cout, dtype=torch.float64)]).view(-1, bm, Cout).sum(1)

    def check_partials(z, part, tile):
        bm = ops.TILE_VARIANTS[tile - 10 if tile > 30 else tile][0]
        assert part.shape == ((M + bm - 1) // bm, 2, Cout) and part.is_contiguous(), tile     # one row per row tile
        zd = z.reshape(M, Cout).double().cpu()
        pc = part.double().cpu()
        assert ((pc[:, 0] - tile_sums(zd, bm)).abs() <= 2e-6 * tile_sums(zd.abs(), bm) + 1e-30).all(), tile
        assert ((pc[:, 1] - tile_sums(zd * zd, bm)).abs() <= 2e-6 * tile_sums(zd * zd, bm) + 1e-30).all(), tile

    first = None
    for tile in [t for t in range(21, 28) if Cout % ops.TILE_VARIANTS[t][1] == 0]:
        z, part = ops.conv2d_stats(x, prm, tile=tile)
        assert torch.equal(z, z_ref), tile
        check_partials(z, part, tile)
        z2, part2 = ops.conv2d_stats(x, prm, tile=tile)
        assert torch.equal(part, part2), tile              # fixed order: reproducible
        if first is None:
            first = part.clone()
    # split-K variants: the remainder tiles add K slices in a different grouping -> fp32 rounding of the regrouped sum
    # (|d| <= 1e-5 max|z| at K <= 2304); reproducible from launch to launch; the partial sums are those of the stored values
    for tile in [t for t in ops.SPLITK_TILES if Cout % ops.TILE_VARIANTS[t - 10][1] == 0]:
        z, part = ops.conv2d_stats(x, prm, tile=tile)
        assert (z - z_ref).abs().max() <= 1e-5 * z_ref.abs().max(), tile
        z2, part2 = ops.conv2d_stats(x, prm, tile=tile)
        assert torch.equal(z, z2) and torch.equal(part, part2), tile
        check_partials(z, part, tile)
    zr = z_ref.reshape(M, Cout).double().cpu()
    rm, rv = _rand(Cout, seed=5), _rand(Cout, seed=6, lo=0.5, hi=1.5)
    rm1, rv1, rm2, rv2 = rm.to(dev), rv.to(dev), rm.to(dev), rv.to(dev)
    mean, invstd = T.bn_stats_partials(first, M, 1e-5, 0.1, rm1, rv1)
    mean0, invstd0 = T.bn_stats(z_ref.reshape(M, Cout), 1e-5, 0.1, rm2, rv2)
    assert torch.allclose(mean, mean0, rtol=1e-5, atol=1e-7) and torch.allclose(invstd, invstd0, rtol=1e-5)
    assert torch.allclose(rm1, rm2, rtol=1e-5, atol=1e-7) and torch.allclose(rv1, rv2, rtol=1e-5, atol=1e-7)
    mu = zr.mean(0)
    assert torch.allclose(mean.double().cpu(), mu, rtol=1e-5, atol=1e-6)
    assert torch.allclose(invstd.double().cpu(), 1 / (zr.var(0, unbiased=False) + 1e-5).sqrt(), rtol=1e-5)
    # one call for statistics + normalisation (+ residual, ReLU, sign mask)
    gamma, beta = _rand(Cout, seed=7, lo=0.5, hi=1.5).to(dev), _rand(Cout, seed=8).to(dev)
    res = _rand(M, Cout, seed=9).to(dev).view_as(z_ref)
    mask1 = torch.empty((M, Cout // 32), dtype=torch.int32, device=dev)
    mask2 = torch.empty_like(mask1)
    rm3, rv3 = rm.to(dev), rv.to(dev)
    y1 = T.bn_apply(z_ref, mean, invstd, gamma, beta, torch.empty_like(z_ref), residual=res, relu=True, mask=mask1)
    y2, mean2, invstd2 = T.bn_fwd_partials(z_ref, first, gamma, beta, torch.empty_like(z_ref), 1e-5, 0.1, rm3, rv3, residual=res,
                                           relu=True, mask=mask2)
    assert torch.equal(y1, y2) and torch.equal(mask1, mask2) and torch.equal(mean, mean2) and torch.equal(invstd, invstd2)
    assert torch.equal(rm1, rm3) and torch.equal(rv1, rv3)


BNBWD_CASES = [  # N, H, W, Cin, Cout, k, p, d, residual, relu
    (2, 51, 51, 256, 64, 1, 0, 1, False, True),       # dgrad of a bottleneck's c3 into bn2
    (2, 51, 51, 64, 64, 3, 1, 1, False, True),        # dgrad of c2 into bn1 (M = 5202: ragged last row group)
    (2, 26, 26, 128, 512, 1, 0, 1, True, True),       # dgrad of the next block's c1 + residual-branch gradient into bn3
    (2, 26, 26, 256, 256, 3, 2, 2, False, True),      # dilated
    (1, 5, 3, 64, 1024, 1, 0, 1, True, False),        # a BatchNorm without ReLU (no mask); M = 15
    (8, 51, 51, 256, 256, 3, 2, 2, False, True),      # training shape: remainder tiles split along K
    (8, 51, 51, 1024, 256, 1, 0, 1, True, True),
]


@pytest.mark.parametrize("case", BNBWD_CASES)
def test_input_gradient_conv_with_batchnorm_backward_in_the_epilogue(hip_lib, dev, case):
    """pemp_bn_apply_mask_f32 + pemp_conv2d_bnbwd_nhwc_f32 + pemp_bn_bwd_partials_f32 against the unfused chain
    (pemp_conv2d_nhwc_f32 -> pemp_bn_bwd_f32): the sign bits are exactly y > 0, the masked gradient g is bit-identical
    (every tile), the partial sums are those of g and g * xhat (fp32 sums of 32 terms), and dz / dgamma / dbeta agree to
    fp32 rounding of the reductions."""
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, p, d, with_res, relu = case
    M = N * H * W
    # forward of the BatchNorm this gradient belongs to
    z = _nhwc(_rand(N, Cout, H, W, seed=11) * 2 + 0.3).to(dev)
    res_f = _nhwc(_rand(N, Cout, H, W, seed=12)).to(dev)
    gamma, beta = _rand(Cout, seed=13, lo=0.5, hi=1.5).to(dev), _rand(Cout, seed=14).to(dev)
    mean, invstd = T.bn_stats(z.view(M, Cout))
    mask = torch.empty((M, Cout // 32), dtype=torch.int32, device=dev) if relu else None
    y = T.bn_apply(z, mean, invstd, gamma, beta, torch.empty_like(z), residual=res_f, relu=relu, mask=mask)
    y_plain = T.bn_apply(z, mean, invstd, gamma, beta, torch.empty_like(z), residual=res_f, relu=relu)
    assert torch.equal(y, y_plain)
    if relu:
        bits = (y.view(M, Cout // 32, 32) > 0).to(torch.int64) << torch.arange(32, device=dev)
        want = bits.sum(-1)
        want = torch.where(want >= 2 ** 31, want - 2 ** 32, want).to(torch.int32)
        assert torch.equal(mask, want)
    # the conv that produces the gradient at the BatchNorm's output
    x = _nhwc(_rand(N, Cin, H, W, seed=1)).to(dev)
    w = _rand(Cout, Cin, k, k, seed=2, lo=-0.1, hi=0.1)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, Cin, Cout, k, k, 1, p, d, kpad, False, False)
    add = _nhwc(_rand(N, Cout, H, W, seed=3)).to(dev) if with_res else None
    dy = ops.conv2d(x, prm, residual=add, tile=13)
    g_ref = torch.where(y > 0, dy, torch.zeros_like(dy)) if relu else dy
    bn = dict(z=z, mean=mean, invstd=invstd, mask=mask)
    xhat = ((z.view(M, Cout) - mean) * invstd).double().cpu()
    gd = g_ref.view(M, Cout).double().cpu()
    def tile_sums(t, bm):
        pad = (-M) % bm
        return torch.cat([t, torch.zeros(pad, Cout, dtype=torch.float64)]).view(-1, bm, Cout).sum(1)

    def check_partials(gv, part, tile):
        bm = ops.TILE_VARIANTS[tile - 10 if tile > 30 else tile][0]
        assert part.shape == ((M + bm - 1) // bm, 2, Cout), tile
        gs = gv.view(M, Cout).double().cpu()
        pc = part.double().cpu()
        assert ((pc[:, 0] - tile_sums(gs, bm)).abs() <= 2e-6 * tile_sums(gs.abs(), bm) + 1e-30).all(), tile
        assert ((pc[:, 1] - tile_sums(gs * xhat, bm)).abs() <= 4e-6 * tile_sums((gs * xhat).abs(), bm) + 1e-30).all(), tile

    first = None
    for tile in [t for t in range(21, 28) if Cout % ops.TILE_VARIANTS[t][1] == 0]:
        g, part = ops.conv2d_bnbwd(x, prm, bn, residual=add, tile=tile)
        assert torch.equal(g, g_ref), tile
        check_partials(g, part, tile)
        if first is None:
            first = part.clone()
    for tile in [t for t in ops.SPLITK_TILES if Cout % ops.TILE_VARIANTS[t - 10][1] == 0]:      # split-K variants (see above)
        g, part = ops.conv2d_bnbwd(x, prm, bn, residual=add, tile=tile)
        assert (g - g_ref).abs().max() <= 1e-5 * dy.abs().max(), tile
        check_partials(g, part, tile)
    dz = torch.empty_like(z)
    dgamma, dbeta = T.bn_bwd_partials(g_ref, z, mean, invstd, gamma, first, dz)
    dz0, gout0 = torch.empty_like(z), torch.empty_like(z)
    dgamma0, dbeta0 = T.bn_bwd(dy, y, z, mean, invstd, gamma, dz0, gout=gout0, relu=relu)
    assert torch.equal(gout0, g_ref)
    if relu:                                    # the sign bits standing in for y: the same results, y never read
        dz1, gout1 = torch.empty_like(z), torch.empty_like(z)
        dgamma1, dbeta1 = T.bn_bwd(dy, None, z, mean, invstd, gamma, dz1, gout=gout1, relu=True, mask=mask)
        assert torch.equal(dz1, dz0) and torch.equal(gout1, gout0) and torch.equal(dgamma1, dgamma0) and torch.equal(dbeta1, dbeta0)
    scale = gd.abs().sum(0).float().to(dev)
    assert ((dbeta - dbeta0).abs() <= 1e-6 * scale + 1e-30).all() and ((dgamma - dgamma0).abs() <= 4e-6 * scale + 1e-30).all()
    assert torch.allclose(dz, dz0, rtol=1e-4, atol=1e-5), (dz - dz0).abs().max()


def test_conv_with_batch_statistics_refuses_what_it_cannot_run(hip_lib, dev):
    from pemp_amd import _lib, ops
    x = torch.zeros(1, 8, 8, 64, device=dev)
    w = torch.zeros(64, 64, 1, 1, device=dev)
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, None, None, 64, 64, 1, 1, 1, 0, 1, kpad, False, False)
    with pytest.raises(_lib.PempHipError, match="tile must be 0, 21..27 or 31..37"):
        ops.conv2d_stats(x, prm, tile=13)


def test_relu_bias_bwd_and_pools(hip_lib, dev):
    from pemp_amd import ops, train_ops as T
    M, C = 1000, 256
    pre = _rand(M, C, seed=1).requires_grad_()
    b = _rand(C, seed=2).requires_grad_()
    y = F.relu(pre + b)
    dy, add = _rand(M, C, seed=3), _rand(M, C, seed=4)
    y.backward(dy + add)
    g = torch.empty(M, C, device=dev)
    db = T.relu_bias_bwd(dy.to(dev), y.detach().to(dev), g, add=add.to(dev))
    assert torch.allclose(g.cpu(), pre.grad, rtol=1e-6, atol=1e-7) and torch.allclose(db.cpu(), b.grad, rtol=1e-4, atol=1e-4)
    # max pool backward (ceil mode, overlapping windows, ties)
    x = (_rand(2, 8, 49, 49, seed=5) * 4).round().requires_grad_()      # many exact ties
    yp = F.max_pool2d(x, 3, 2, 1, ceil_mode=True)
    gp = _rand(*yp.shape, seed=6)
    yp.backward(gp)
    dx = T.maxpool_bwd(_nhwc(x.detach()).to(dev), _nhwc(gp).to(dev), 3, 2, 1)
    assert torch.allclose(_nchw(dx.cpu()), x.grad, rtol=1e-6, atol=1e-6)
    # index-recording form used by the training engines: same output, same gradient (bit-exact), for the
    # ResNet stem pool (3/2/1 ceil) and the VGG stride-1 pool (3/1/1)
    for s, ceil in ((2, True), (1, False)):
        xs = (_rand(2, 8, 49, 49, seed=9 + s) * 4).round().requires_grad_()
        ys = F.max_pool2d(xs, 3, s, 1, ceil_mode=ceil)
        gs = _rand(*ys.shape, seed=12 + s)
        ys.backward(gs)
        xd = _nhwc(xs.detach()).to(dev)
        yi, idx = T.maxpool_idx(xd, 3, s, 1, ceil_mode=ceil)
        assert torch.equal(_nchw(yi.cpu()), ys.detach()) and idx.dtype == torch.uint8 and int(idx.max()) <= 8
        dxi = T.maxpool_idx_bwd(idx, _nhwc(gs).to(dev), xd.shape[1:3], 3, s, 1)
        assert torch.equal(dxi, T.maxpool_bwd(xd, _nhwc(gs).to(dev), 3, s, 1))
        assert torch.allclose(_nchw(dxi.cpu()), xs.grad, rtol=1e-6, atol=1e-6)
    # global average pool backward
    xg = _rand(3, 64, 5, 7, seed=7).requires_grad_()
    v = _rand(3, 64, seed=8)
    F.adaptive_avg_pool2d(xg, (1, 1)).flatten(1).backward(v)
    dst = torch.zeros(3, 5, 7, 64, device=dev)
    T.gap_bwd_add(v.to(dev), dst)
    assert torch.allclose(_nchw(dst.cpu()), xg.grad, rtol=1e-6, atol=1e-7)
    src = _rand(2, 4, 5, 8, seed=9)
    sc = T.scatter_strided(src.to(dev), (7, 9), 2).cpu()
    assert torch.equal(sc[:, ::2, ::2], src) and sc.sum() == src.sum()


def test_sgd_clip_step_matches_torch(hip_lib, dev):
    from pemp_amd import train_ops as T
    n = 100003
    p0, g1, g2 = _rand(n, seed=1), _rand(n, seed=2) * 0.01, _rand(n, seed=3) * 3
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    pd, buf = p0.clone().to(dev), torch.zeros(n, device=dev)
    for step, g in enumerate((g1, g2)):
        ref.grad = g.clone()
        tot = torch.nn.utils.clip_grad_norm_([ref], 1.1)
        opt.step()
        norm = T.sgd_clip_step(pd, g.to(dev), buf, 1.1, 1e-3, 0.9, 5e-4, first_step=(step == 0))
        assert abs(norm.item() - tot.item()) <= 1e-5 * tot.item()
        assert torch.allclose(pd.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)
    # grad_scale = 1/world on summed gradients equals averaging first
    pa, pb = p0.clone().to(dev), p0.clone().to(dev)
    T.sgd_clip_step(pa, (g2 * 4).to(dev), torch.zeros(n, device=dev), 1.1, 1e-3, 0.9, 5e-4, True, grad_scale=0.25)
    T.sgd_clip_step(pb, g2.to(dev), torch.zeros(n, device=dev), 1.1, 1e-3, 0.9, 5e-4, True)
    assert torch.allclose(pa.cpu(), pb.cpu(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,S,p,c,h,w,H,W,Ho,Wo", [(2, 2, 3, 512, 9, 11, 70, 85, 50, 61), (1, 1, 2, 256, 7, 7, 50, 50, 33, 40),
                                                   (2, 1, 0, 512, 6, 5, 41, 37, 41, 37), (1, 5, 3, 128, 5, 6, 40, 47, 40, 47),
                                                   (2, 1, 5, 512, 7, 6, 41, 37, 41, 37), (1, 2, 8, 256, 5, 6, 40, 47, 33, 40)])   # protos 5..8: MAXJ = 16
def test_head_backward_matches_autograd(hip_lib, dev, B, S, p, c, h, w, H, W, Ho, Wo):
    """pemp_head_bwd_f32 vs torch autograd through the torch restatement of the head (CPU): against its float32 evaluation at the
    tolerance two float32 evaluations of this head can be held to (2e-4 / 5e-4 of the maximum), and against its float64
    evaluation with the kernel's own winning prototypes replayed at 1e-5 in L2."""
    from pemp_amd import ops, train_ops as T
    from tests.util import head_loss
    feat = (_rand(B * S + B, h, w, c, seed=1) * 2).requires_grad_()
    m = (_rand(B, S, 1, H, W, seed=3) > 0.1).float()
    mask = torch.cat((m, 1 - m), dim=2)
    ctr = _rand(c, 2 * p, seed=4, lo=0, hi=1).requires_grad_() if p > 0 else None
    tgt = (_rand(B, Ho, Wo, seed=5) > 0.2).long()
    tgt[0, :3] = 255
    loss, _ = head_loss(feat, mask, tgt, ctr, B, S, 1, p, 20.0, (Ho, Wo))
    grads = torch.autograd.grad(loss, [feat] + ([ctr] if p > 0 else []))
    fd, md, td = feat.detach().to(dev), mask.reshape(B * S, 2, H, W).to(dev), tgt.to(dev)
    ws = {}
    sup, qry = fd[:B * S], fd[B * S:]
    if p > 0:
        pro = ops.mpm_protos(sup, md, ctr.detach().to(dev), B, S, p, ws_cache=ws)
        key = ("mpm", B, S, h, w, c, p)
    else:
        pro = ops.masked_avg_pool(sup, md, B, S, full_res=False, ws_cache=ws)
        key = ("map", B, S, h, w, c)
    pred = ops.cosine_proto_max(qry, pro, 20.0)
    _, stats, _ = ops.eval_tail(pred, td, ws_cache=ws)
    got_loss = (stats[:, 0].sum() / stats[:, 1].sum()).item()
    assert abs(got_loss - loss.item()) < 1e-5
    dfeat = torch.empty_like(fd)
    dctr = T.head_bwd(sup, qry, md, ctr.detach().to(dev) if p > 0 else None, ws[key], pro, pred, td, stats, dfeat, B, S, p,
                      20.0, ws_cache=ws)
    scale = grads[0].abs().max().item()
    assert (dfeat.cpu() - grads[0]).abs().max().item() < 2e-4 * scale + 1e-9, (dfeat.cpu() - grads[0]).abs().max().item() / scale
    if p > 0:
        cs = grads[1].abs().max().item()
        assert (dctr.cpu() - grads[1]).abs().max().item() < 5e-4 * cs + 1e-9, (dctr.cpu() - grads[1]).abs().max().item() / cs
    # the same head in float64 with the winners the kernel itself took (frozen decision): rounding only
    win = None
    if p > 0:
        from pemp_amd import _lib
        n = h * w
        nbytes = _lib.load().pemp_head_bwd_workspace_bytes(B, S, n, c, p)
        wk = ws[("head_bwd", B, S, h, w, c, p)][nbytes - B * 2 * n * 4:nbytes].view(torch.int32).view(B, 2, h, w).cpu().long()
        win = torch.stack((wk[:, 1] - p, wk[:, 0]), dim=1)
    f64 = feat.detach().double().requires_grad_()
    c64 = ctr.detach().double().requires_grad_() if p > 0 else None
    l64, _ = head_loss(f64, mask.double(), tgt, c64, B, S, 1, p, 20.0, (Ho, Wo), win=win)
    g64 = torch.autograd.grad(l64, [f64] + ([c64] if p > 0 else []))
    rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
    e = [rel(dfeat.cpu(), g64[0]), rel(grads[0], g64[0])] + ([rel(dctr.cpu(), g64[1]), rel(grads[1], g64[1])] if p > 0 else [])
    print(f"head backward B{B} S{S} p{p} c{c}: relative L2 error vs float64  dfeat hip {e[0]:.2e} torch-fp32 {e[1]:.2e}"
          + (f"   dctr hip {e[2]:.2e} torch-fp32 {e[3]:.2e}" if p > 0 else ""))
    assert abs(got_loss - l64.item()) < 2e-6
    # measured 3e-7 .. 2e-6 (torch float32 on the reference's formulation: 7e-6 .. 3.4e-5 -- the rounding of its squared
    # distances, which the kernels' shift-invariant logits do not have, csrc/head_common.h)
    assert max(e[0::2]) < 1e-5, e


def test_cm_linear_bias_and_backward_match_autograd(hip_lib, dev):
    """ResNetCM.comm's small linear algebra (episode mean, Linear(2C->2), per-image conv bias of the two constant
    channels) and its backward vs torch autograd on the reference's formulation (backbones.py:213-221)."""
    from pemp_amd import ops, train_ops as T
    torch.manual_seed(3)
    N, G, C, co = 9, 3, 64, 48
    group = torch.tensor([0, 0, 1, 2, 1, 0, 2, 2, 1], dtype=torch.int32, device=dev)       # uneven episodes, any order
    stat = torch.randn(N, 2, C, device=dev)
    W = torch.randn(2, 2 * C, device=dev, requires_grad=True)
    b = torch.randn(2, device=dev, requires_grad=True)
    wfull = torch.randn(co, C + 2, device=dev, requires_grad=True)                          # [Cout, C+2]: last two = comm columns
    alpha, base = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    st = stat.clone().requires_grad_(True)
    g64 = group.long()
    cnt = torch.bincount(g64, minlength=G).float()[:, None]
    agg_ref = torch.zeros(G, 2 * C, device=dev).index_add(0, g64, st.view(N, -1)) / cnt
    feat_ref = torch.addmm(b, agg_ref, W.t())
    bias_ref = feat_ref[g64] @ wfull[:, C:].t()
    agg, feat = ops.cm_linear(stat, group, W.detach(), b.detach(), G)
    assert torch.allclose(agg, agg_ref.detach(), atol=1e-6) and torch.allclose(feat, feat_ref.detach(), atol=1e-5)
    wext = wfull.detach()[:, C:]
    assert torch.allclose(ops.cm_bias(feat, group, wext), bias_ref.detach(), atol=1e-5)
    assert torch.allclose(ops.cm_bias(feat, group, wext, alpha=alpha, base=base), base + alpha * bias_ref.detach(), atol=1e-5)
    colsum = torch.randn(N, co, device=dev)
    (bias_ref * colsum).sum().backward()
    dwfull = torch.zeros(co, C + 2, device=dev)
    dfi = torch.empty(N, 2, device=dev)
    T.cm_bias_bwd(colsum, feat, group, wext, dwfull[:, C:], dfi, accumulate=False)
    assert torch.allclose(dwfull[:, C:], wfull.grad[:, C:], rtol=1e-4, atol=1e-5) and not dwfull[:, :C].any()
    dW, db = torch.empty_like(W), torch.empty_like(b)
    dstat = T.cm_linear_bwd(dfi, group, agg, W.detach(), dW, db)
    assert torch.allclose(dW, W.grad, rtol=1e-4, atol=1e-5) and torch.allclose(db, b.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(dstat.view(N, 2, C), st.grad, rtol=1e-4, atol=1e-6)
    dfi2 = dfi.clone()
    T.cm_bias_bwd(colsum, feat, group, wext, dwfull[:, C:], dfi2, accumulate=True)
    assert torch.allclose(dfi2, 2 * dfi, rtol=1e-6)


@pytest.mark.parametrize("case", [  # N, H, W, Cin, Cout, k, p, d, residual, relu
    (8, 51, 51, 256, 256, 3, 2, 2, False, True), (8, 51, 51, 256, 1024, 1, 0, 1, True, True), (8, 51, 51, 1024, 256, 1, 0, 1, False, False),
    (2, 51, 51, 128, 128, 3, 1, 1, True, False), (1, 9, 7, 64, 64, 1, 0, 1, False, True)])
def test_split_k_conv_variants_match_the_unsplit_conv(hip_lib, dev, case):
    """pemp_conv2d_splitk_nhwc_f32 (tile ids 31..37: the last round of tiles split along K, partial tiles through the
    uncached workspace, fixed-order fix-up by the last block to arrive) against the unsplit variants: scale / shift /
    residual / ReLU epilogue included; |d| <= 1e-5 max|y| (regrouped fp32 sum), bit-identical from launch to launch, and
    bit-identical to the unsplit variant where the geometry leaves nothing to split."""
    from pemp_amd import ops
    N, H, W, Cin, Cout, k, p, d, with_res, relu = case
    x = _nhwc(_rand(N, Cin, H, W, seed=1)).to(dev)
    w = _rand(Cout, Cin, k, k, seed=2, lo=-0.1, hi=0.1)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, _rand(Cout, seed=3, lo=0.5, hi=1.5).to(dev), _rand(Cout, seed=4).to(dev), Cin, Cout, k, k, 1, p, d,
                         kpad, False, relu)
    res = _nhwc(_rand(N, Cout, H, W, seed=5)).to(dev) if with_res else None
    ref = ops.conv2d(x, prm, residual=res, tile=13)
    lib = hip_lib
    M = N * H * W
    for tile in [t for t in ops.SPLITK_TILES if Cout % ops.TILE_VARIANTS[t - 10][1] == 0]:
        y1 = ops.conv2d(x, prm, residual=res, tile=tile)
        y2 = ops.conv2d(x, prm, residual=res, tile=tile)
        assert torch.equal(y1, y2), tile
        assert (y1 - ref).abs().max() <= 1e-5 * ref.abs().max(), tile
        desc = ops.ConvDesc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, p, d, 0, kpad, 0, tile)
        if lib.pemp_conv2d_splitk_workspace_bytes(ops.C.byref(desc)) == 0:       # nothing to split at this geometry
            assert torch.equal(y1, ref), tile
    if M >= 8 * 51 * 51:
        desc = ops.ConvDesc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, p, d, 0, kpad, 0, 31)
        assert lib.pemp_conv2d_splitk_workspace_bytes(ops.C.byref(desc)) > 0       # the training shapes do split


@pytest.mark.parametrize("tile", [31, 32, 34, 35, 36, 37])
def test_split_k_hand_off_is_complete_and_stable_under_uneven_load(hip_lib, dev, tile):
    """The cross-XCD hand-off of the split-K variants (partial tiles through uncached memory, `s_waitcnt vmcnt(0)` + block
    barrier + one agent-scope atomic add per block, the last block to arrive reads every piece back; conv_dma2.hip) for
    remainder-tile counts 1 .. 128: every output word of every launch equals the first launch's (a piece read before it
    had landed would change words of that tile), and the result is the unsplit conv's up to the rounding of the regrouped
    sum -- while a second stream keeps the memory system busy with a large copy and idle gaps (uneven load: the
    condition under which a missing release / acquire shows, MI355X_MICROARCH.md "Test every hand-off")."""
    from pemp_amd import ops
    bm, bn = ops.TILE_VARIANTS[tile - 10]
    Cin, Cout = 256, bn                                     # one column of tiles: the tile count is ceil(M / bm)
    w = _rand(Cout, Cin, 1, 1, seed=2, lo=-0.1, hi=0.1)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, Cin, Cout, 1, 1, 1, 0, 1, kpad, False, False)
    big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)       # 256 MB each: past the Infinity Cache
    side = torch.cuda.Stream(device=dev)
    for rem in (1, 2, 5, 31, 64, 100, 128):
        rows = (256 + rem) * bm - 5                         # ragged last tile
        H = 64
        Wd = (rows + H - 1) // H
        x = torch.randn(1, H, Wd, Cin, device=dev, generator=torch.Generator(device=dev).manual_seed(rem))
        desc = ops.ConvDesc(1, H, Wd, Cin, Cin, H, Wd, Cout, Cout, 1, 1, 1, 0, 1, 0, kpad, 0, tile)
        split = hip_lib.pemp_conv2d_splitk_workspace_bytes(ops.C.byref(desc)) > 0
        T_ = (H * Wd + bm - 1) // bm
        assert split == (1 <= T_ % 256 <= 128), (rem, T_)
        ref = ops.conv2d(x, prm, tile=tile - 10)
        first = ops.conv2d(x, prm, tile=tile).clone()
        assert (first - ref).abs().max() <= 1e-5 * ref.abs().max(), (tile, rem)
        side.wait_stream(torch.cuda.current_stream())
        for rep in range(6):
            with torch.cuda.stream(side):                   # bursts of traffic beside every other launch
                if rep % 2 == 0:
                    big_b.copy_(big_a)
            y = ops.conv2d(x, prm, tile=tile)
            assert torch.equal(y, first), (tile, rem, rep)
        torch.cuda.current_stream().wait_stream(side)


def test_dgrad_mirror_of_a_flat_buffer_matches_the_per_layer_transpose(hip_lib, dev):
    """pemp_dgrad_mirror_f32: the input-gradient weights of every conv layer of a flat parameter buffer in one launch ==
    train_ops.dgrad_weight (flip + transpose in torch) layer by layer; ragged channel counts (3, 66), 1x1 / 3x3 / 7x7,
    bytes between the layers (BatchNorm parameters) left untouched."""
    from pemp_amd import train_ops as T
    shapes = [(64, 3, 7), (64, 64, 1), (256, 66, 1), (128, 128, 3), (512, 1280, 1), (96, 160, 3)]      # (Cout, Cin, k)
    g = torch.Generator().manual_seed(5)
    chunks, layers, off = [], [], 0
    for co, ci, k in shapes:
        gap = torch.rand(12, generator=g)                                   # something that is not a conv weight
        chunks.append(gap)
        off += 12
        w = torch.rand(co, k * k * ci, generator=g)
        chunks.append(w.reshape(-1))
        layers.append((off, co, k * k, ci))
        off += w.numel()
    flat = torch.cat(chunks).to(dev)
    mirror = torch.full_like(flat, -7.0)
    table, tiles = T.dgrad_mirror_table(layers, dev)
    T.dgrad_mirror(flat, mirror, table, tiles)
    pos = 0
    for (off, co, taps, ci), (_, _, k) in zip(layers, shapes):
        assert bool((mirror[pos:off] == -7.0).all())
        n = co * taps * ci
        ref = T.dgrad_weight(flat[off:off + n].view(co, taps * ci), k, k)
        assert torch.equal(mirror[off:off + n].view(ci, taps * co), ref), (co, ci, k)
        pos = off + n


def test_concurrent_stream_runs_beside_the_current_one(hip_lib, dev):
    """ops.concurrent_stream: whatever number of streams the process has created before (HIP deals them round-robin to a few
    hardware queues; the n-th one can share the current stream's queue), the stream it returns overlaps with the current
    stream: two 300 us idle kernels, one on each, finish in well under 600 us."""
    import ctypes as C
    from pemp_amd import ops, _lib
    cur = torch.cuda.current_stream(dev)
    for created_before in range(5):
        junk = [torch.cuda.Stream(device=dev) for _ in range(created_before)]
        s = ops.concurrent_stream(dev)
        s.wait_stream(cur)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        _lib.check(hip_lib.pemp_spin_us(300, C.c_void_p(cur.cuda_stream)), "spin")
        _lib.check(hip_lib.pemp_spin_us(300, C.c_void_p(s.cuda_stream)), "spin")
        cur.wait_stream(s)
        e1.record(cur)
        e1.synchronize()
        assert e0.elapsed_time(e1) < 480, (created_before, e0.elapsed_time(e1))
        del junk


def test_split_k_with_a_padding_value_matches_the_unsplit_conv(hip_lib, dev):
    """pemp_conv2d_padv_splitk_nhwc_f32 (the dilated ASPPV2 branch convs of a one-episode evaluation step: BatchNorm folded in
    front of a zero-padded conv, out-of-image taps read a per-channel value; 5202 rows = 82 tiles, all of them split along K):
    every split-K variant against the unsplit padding-value conv, |d| <= 1e-5 max|y|, stable from launch to launch."""
    from pemp_amd import ops
    N, H, W, Cin, Cout = 2, 51, 51, 256, 256
    for d in (6, 12):
        buf = torch.empty(N * H * W + 4, Cin, device=dev)                      # the padding vector sits behind the tensor
        buf.copy_(_rand(N * H * W + 4, Cin, seed=11).to(dev))
        x, pv = buf[:N * H * W].view(N, H, W, Cin), buf[N * H * W]
        w = _rand(Cout, Cin, 3, 3, seed=2, lo=-0.1, hi=0.1)
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, _rand(Cout, seed=4).to(dev), Cin, Cout, 3, 3, 1, d, d, kpad, False, True)
        ref = ops.conv2d(x, prm, pad_value=pv, tile=27)
        plain = ops.conv2d(x, prm, tile=27)
        assert (ref - plain).abs().max() > 1e-2                                 # the padding value really enters
        for tile in [t for t in ops.SPLITK_TILES if Cout % ops.TILE_VARIANTS[t - 10][1] == 0]:
            y1 = ops.conv2d(x, prm, pad_value=pv, tile=tile).clone()
            y2 = ops.conv2d(x, prm, pad_value=pv, tile=tile)
            assert torch.equal(y1, y2), (d, tile)
            assert (y1 - ref).abs().max() <= 1e-5 * ref.abs().max(), (d, tile)
