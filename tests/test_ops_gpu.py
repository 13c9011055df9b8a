"""Parity of every C-ABI entry point against plain torch fp32 ops on the CPU (the op-level oracle).

Tolerances: conv outputs are sums of K exact fp32 products in a different order than oneDNN's,
so |diff| <= 2e-5 * (1 + |ref|) * sqrt(K/64) is the bound used; streaming kernels 1e-5 relative.
"""
import numpy as np
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


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, p, d, tile
    (2, 13, 13, 64, 64, 1, 1, 0, 1, 3),
    (2, 25, 25, 256, 128, 1, 2, 0, 1, 3),
    (1, 51, 51, 256, 256, 3, 1, 2, 2, 3),
    (2, 51, 51, 256, 1024, 1, 1, 0, 1, 1),
    (1, 51, 51, 256, 256, 3, 1, 18, 18, 2),
    (2, 13, 13, 256, 256, 3, 1, 6, 6, 0),
    (1, 17, 23, 32, 128, 3, 1, 1, 1, 1),
    (3, 9, 7, 128, 64, 3, 2, 1, 1, 2),
    (1, 30, 30, 1024, 256, 1, 1, 0, 1, 0),
]


@pytest.mark.parametrize("dma", [0, 10, 14, 15, 16, 17, 20, 24, 25, 26, 27])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_generic(hip_lib, dev, case, dma):
    from pemp_amd import ops
    N, H, W, Cin, Cout, k, s, p, d, tile = case
    if dma % 10 >= 4:                   # 8-wave LDS-DMA blocks: 128x128 (needs Cout % 128 == 0) / 128x64
        need = {4: 128, 5: 64, 6: 128, 7: 256}[dma % 10]    # x6 / x7: 256x128 / 256x256 tiles (98 / 131 KB of LDS)
        tile = dma if Cout % need == 0 else dma - dma % 10 + 5
    else:                               # 10: conv_dma.hip, 20: conv_dma2.hip, 4-wave blocks
        tile = tile + dma if tile else (dma + 3 if dma else 0)
    x = _rand(N, Cin, H, W, seed=1)
    w = _rand(Cout, Cin, k, k, seed=2) * (1.0 / (Cin * k * k) ** 0.5)
    scale = _rand(Cout, seed=3, lo=0.5, hi=1.5)
    shift = _rand(Cout, seed=4)
    ref = F.conv2d(x, w, None, s, p, d) * scale[None, :, None, None] + shift[None, :, None, None]
    res = _rand(*ref.shape, seed=5)
    ref_full = F.relu(ref + res)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, scale.to(dev), shift.to(dev), Cin, Cout, k, k, s, p, d, kpad, False, True)
    y = ops.conv2d(_nhwc(x).to(dev), prm, residual=_nhwc(res).to(dev), tile=tile)
    torch.cuda.synchronize()
    got = _nchw(y.cpu())
    tol = 2e-5 * (Cin * k * k / 64) ** 0.5
    err = ((got - ref_full).abs() / (1 + ref_full.abs())).max().item()
    assert err < tol, f"{case}: err {err} tol {tol}"
    # no relu / no residual / no affine
    prm2 = ops.ConvParams(packed, None, None, Cin, Cout, k, k, s, p, d, kpad, False, False)
    y2 = _nchw(ops.conv2d(_nhwc(x).to(dev), prm2, tile=tile).cpu())
    ref2 = F.conv2d(x, w, None, s, p, d)
    assert ((y2 - ref2).abs() / (1 + ref2.abs())).max().item() < tol


def test_conv2d_channel_slices_and_per_image_shift(hip_lib, dev):
    from pemp_amd import ops
    N, H, W, Cin, Cout = 3, 11, 11, 64, 128
    wide_in = _rand(N, H, W, 96, seed=7)            # read channels 32..96 of a 96-wide buffer
    w = _rand(Cout, Cin, 1, 1, seed=8) * 0.1
    shift = _rand(N, Cout, seed=9)
    ref = F.conv2d(wide_in[..., 32:].permute(0, 3, 1, 2), w) + shift[:, :, None, None]
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, Cin, Cout, 1, 1, 1, 0, 1, kpad, False, False)
    xin = wide_in.to(dev)
    wide_out = torch.full((N, H, W, 320), -7.0, device=dev)
    ops.conv2d(xin[..., 32:], prm, out=wide_out[..., 64:192], shift_override=shift.to(dev).contiguous(), per_image_shift=True)
    torch.cuda.synchronize()
    got = wide_out.cpu()
    assert (got[..., :64] == -7).all() and (got[..., 192:] == -7).all()
    assert torch.allclose(got[..., 64:192].permute(0, 3, 1, 2), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,d", [(2, 51, 51, 64, 256, 3, 2), (3, 37, 45, 96, 256, 1, 1), (2, 101, 101, 64, 64, 3, 1),
                                                 (1, 72, 73, 32, 1024, 1, 1)])
def test_sixteen_row_and_hybrid_tiles_with_every_epilogue_option(hip_lib, dev, N, H, W, Cin, Cout, k, d):
    """Tile ids 28 (16-row wave tiles on v_mfma_f32_16x16x4_f32) and 29 (hybrid: whole rounds on the 64 x 64 tile + the remaining rows
    on 16-row tiles, one grid) against tile 23, bit for bit, with everything their own epilogue / row-tile offset has to get
    right: input and output as channel slices of wider buffers, a residual with its own stride, per-channel scale, a PER-IMAGE
    shift, ReLU, and row counts that end inside a 16-row tile (4995 = 156 x 32 + 3; 5256 = 5120 + 8 x 16 + 8).  The geometries are ones the
    hybrid really splits (asserted)."""
    from pemp_amd import ops
    assert ops.hybrid_rows(N, H, W, Cout) > 0
    xw = _rand(N, H, W, Cin + 32, seed=1).to(dev)
    x = xw[..., 32:]
    w = _rand(Cout, Cin, k, k, seed=2) * (1.0 / (Cin * k * k) ** 0.5)
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    scale = _rand(Cout, seed=3, lo=0.5, hi=1.5).to(dev)
    prm = ops.ConvParams(packed, scale, None, Cin, Cout, k, k, 1, d * (k // 2), d, kpad, False, True)
    shift = _rand(N, Cout, seed=4).to(dev).contiguous()
    resw = _rand(N, H, W, Cout + 64, seed=5).to(dev)
    res = resw[..., 64:]
    outs = {}
    for tile in (23, 28, 29):
        yw = torch.full((N, H, W, Cout + 128), -3.0, device=dev)
        ops.conv2d(x, prm, out=yw[..., 64:64 + Cout], residual=res, shift_override=shift, per_image_shift=True, tile=tile)
        assert bool((yw[..., :64] == -3).all()) and bool((yw[..., 64 + Cout:] == -3).all()), tile
        outs[tile] = yw[..., 64:64 + Cout].clone()
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w, None, 1, d * (k // 2), d) * scale.cpu()[None, :, None, None] + shift.cpu()[:, :, None, None]
    ref = F.relu(ref + res.permute(0, 3, 1, 2).cpu())
    got = outs[23].permute(0, 3, 1, 2).cpu()
    assert ((got - ref).abs() / (1 + ref.abs())).max().item() < 3e-5 * max(1.0, (Cin * k * k / 64) ** 0.5)
    assert torch.equal(outs[28], outs[23]) and torch.equal(outs[29], outs[23])


@pytest.mark.parametrize("Cin,Cout,k,dil,HW", [(64, 64, 3, 2, 13), (64, 128, 3, 6, 19), (256, 256, 3, 18, 51), (64, 64, 1, 1, 9)])
def test_conv2d_pad_value_folds_a_leading_batchnorm(hip_lib, dev, Cin, Cout, k, dil, HW):
    """ASPPV2 (networks/backbones.py:330-357): BN -> conv with the BN OUTPUT zero-padded.  The folded conv reads
    -t/s at out-of-image taps; every kernel variant agrees bit for bit."""
    from pemp_amd import ops, _lib
    N = 2
    x = _rand(N, Cin, HW, HW, seed=1)
    w = _rand(Cout, Cin, k, k, seed=2) * (1.0 / (Cin * k * k) ** 0.5)
    b = _rand(Cout, seed=3)
    s = _rand(Cin, seed=4, lo=0.5, hi=1.5) * torch.where(_rand(Cin, seed=5) > 0, 1.0, -1.0)
    t = _rand(Cin, seed=6)
    pad = dil if k == 3 else 0
    ref = F.relu(F.conv2d(x * s[None, :, None, None] + t[None, :, None, None], w, b, 1, pad, dil))
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, b.to(dev), Cin, Cout, k, k, 1, pad, dil, kpad, False, True)
    q, padv = ops.fold_input_affine(prm, s.to(dev), t.to(dev))
    xd = _nhwc(x).to(dev)
    tol = 3e-5 * (Cin * k * k / 64) ** 0.5
    outs = []
    for tile in (3, 2, 13, 12, 15, 23, 22, 25, 28, 29) + ((11, 14, 16, 21, 24, 26) if Cout % 128 == 0 else ()) + ((17, 27) if Cout % 256 == 0 else ()):
        y = ops.conv2d(xd, q, tile=tile, pad_value=padv if k > 1 else None)
        outs.append(y)
        assert ((_nchw(y.cpu()) - ref).abs() / (1 + ref.abs())).max().item() < tol, tile
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    if k > 1:
        # the buffer-addressed kernels (2x ids) fetch the padding vector through the activations' descriptor: that needs the
        # vector BEHIND the tensor in the same allocation (the engine parks it in spare rows there); with it elsewhere -- the
        # loop above -- those ids hand the launch to conv_dma.hip.  Same results, bit for bit, either way.
        flat = torch.empty(N * HW * HW + 4, Cin, device=dev)
        flat[:N * HW * HW].copy_(xd.view(-1, Cin))
        flat[N * HW * HW + 2].copy_(padv)
        xin, pin = flat[:N * HW * HW].view(N, HW, HW, Cin), flat[N * HW * HW + 2]
        for tile in (23, 22, 25, 28, 29) + ((21, 24, 26) if Cout % 128 == 0 else ()) + ((27,) if Cout % 256 == 0 else ()):
            assert torch.equal(ops.conv2d(xin, q, tile=tile, pad_value=pin), outs[0]), tile
    if k > 1:       # zero padding is NOT the same thing (the border differs), and 1x1 convs reject a pad value
        y0 = ops.conv2d(xd, q, tile=3)
        assert not torch.allclose(_nchw(y0.cpu()), ref, rtol=1e-3, atol=1e-3)
    else:
        with pytest.raises(_lib.PempHipError, match="multi-tap"):
            ops.conv2d(xd, q, tile=3, pad_value=padv)
    assert ops.fold_input_affine(prm, torch.zeros(Cin, device=dev), t.to(dev)) is None        # s == 0: no fold


@pytest.mark.parametrize("cin,k,s,p,H", [(3, 7, 2, 3, 97), (3, 3, 1, 1, 33), (4, 7, 2, 3, 50)])
@pytest.mark.parametrize("tile", [0, 2, 3, 12, 13, 15])
def test_conv2d_stem4(hip_lib, dev, cin, k, s, p, H, tile):
    from pemp_amd import ops
    N, Cout = 2, 64
    x = _rand(N, cin, H, H, seed=11)
    w = _rand(Cout, cin, k, k, seed=12) * 0.2
    ref = F.relu(F.conv2d(x, w, None, s, p))
    x4 = torch.zeros(N, H, H, 4)
    x4[..., :cin] = x.permute(0, 2, 3, 1)
    packed, kpad = ops.pack_conv_weight(w.to(dev), stem4=True)
    prm = ops.ConvParams(packed, None, None, 4, Cout, k, k, s, p, 1, kpad, True, True)
    y = _nchw(ops.conv2d(x4.to(dev), prm, tile=tile).cpu())
    assert torch.allclose(y, ref, rtol=2e-5, atol=2e-5)


def test_conv2d_rejects_bad_arguments(hip_lib, dev):
    from pemp_amd import ops, _lib
    x = torch.zeros(1, 5, 5, 48, device=dev)
    w = torch.zeros(64, 48, 1, 1, device=dev)
    packed, kpad = ops.pack_conv_weight(w)
    prm = ops.ConvParams(packed, None, None, 48, 64, 1, 1, 1, 0, 1, kpad, False, False)
    with pytest.raises(_lib.PempHipError, match="multiple of 32"):
        ops.conv2d(x, prm)
    with pytest.raises(_lib.PempHipError, match="no CPU path"):
        ops.conv2d(x.cpu(), prm)


def test_pack_input_and_pools(hip_lib, dev):
    from pemp_amd import ops
    img = _rand(2, 3, 37, 41, seed=1)
    prior = _rand(2, 1, 37, 41, seed=2)
    x4 = ops.pack_input(img.to(dev), prior.to(dev)).cpu()
    assert torch.equal(x4[..., :3], img.permute(0, 2, 3, 1)) and torch.equal(x4[..., 3], prior[:, 0])
    assert (ops.pack_input(img.to(dev)).cpu()[..., 3] == 0).all()
    x = _rand(2, 64, 49, 49, seed=3)
    for k, s, p, ceil in ((3, 2, 1, True), (3, 2, 1, False), (3, 1, 1, False)):
        ref = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
        got = _nchw(ops.maxpool2d(_nhwc(x).to(dev), k, s, p, ceil_mode=ceil).cpu())
        assert torch.equal(got, ref), (k, s, p, ceil)
    x201 = _rand(1, 8, 201, 201, seed=4)
    assert torch.equal(_nchw(ops.maxpool2d(_nhwc(x201).to(dev), 3, 2, 1, ceil_mode=True).cpu()),
                       F.max_pool2d(x201, 3, 2, 1, ceil_mode=True))
    g = ops.global_avgpool(_nhwc(x).to(dev)).cpu()
    assert torch.allclose(g, x.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)


def test_channel_affine_multi(hip_lib, dev):
    from pemp_amd import ops
    x = _rand(2, 5, 7, 64, seed=1)
    sc = [_rand(64, seed=10 + i) for i in range(4)]
    sh = [_rand(64, seed=20 + i) for i in range(4)]
    outs = [torch.empty(2, 5, 7, 64, device=dev) for _ in range(4)]
    ops.channel_affine_multi(x.to(dev), [s.to(dev) for s in sc], [s.to(dev) for s in sh], outs)
    for i in range(4):
        assert torch.allclose(outs[i].cpu(), x * sc[i] + sh[i], rtol=1e-6, atol=1e-7)   # GPU contracts to one FMA


def _ref_mpm(sup, qry, fg, bg, ctr, p, ret_ind=True):
    from oracle import ref_cpu
    return ref_cpu.mpm(sup, qry, fg, bg, ctr, p, 20, ret_ind)


@pytest.mark.parametrize("B,S,p,c,h,w,H,W", [(1, 1, 3, 512, 13, 13, 97, 97), (2, 5, 3, 512, 9, 11, 70, 85),
                                            (1, 2, 2, 256, 7, 7, 50, 50), (1, 1, 1, 128, 5, 6, 40, 47),
                                            (3, 2, 1, 64, 5, 7, 40, 33), (1, 3, 4, 260, 6, 6, 31, 47),      # c = 260: VALU cosine
                                            (2, 1, 2, 96, 17, 3, 50, 20), (4, 1, 3, 512, 6, 5, 19, 15),
                                            (1, 2, 4, 128, 9, 7, 33, 41),                                    # 2p = 8 on the MFMA path
                                            (1, 1, 3, 512, 1, 1, 9, 9), (2, 2, 3, 512, 3, 5, 20, 33),       # maps smaller than one 16-pixel tile
                                            (1, 2, 5, 512, 9, 7, 33, 41), (2, 1, 6, 256, 7, 9, 50, 47),      # protos 5..8: the MAXJ = 16 instantiation
                                            (1, 1, 8, 128, 13, 13, 97, 97)])
def test_mpm_and_cosine(hip_lib, dev, B, S, p, c, h, w, H, W):
    from pemp_amd import ops
    sup = _rand(B, S, c, h, w, seed=1) * 2
    qry = _rand(B, 1, c, h, w, seed=2) * 2
    m = (_rand(B * S, 1, H, W, seed=3) > 0.2).float()
    mask = torch.cat((m, 1 - m), dim=1)
    ctr = _rand(c, 2 * p, seed=4, lo=0, hi=1)
    mlow = F.interpolate(mask, (h, w), mode="nearest")
    pred_ref, resp_ref, protos_ref = _ref_mpm(sup, qry, mlow[:, 0], mlow[:, 1], ctr, p)
    supn = sup.reshape(B * S, c, h, w).permute(0, 2, 3, 1).contiguous().to(dev)
    qryn = qry.reshape(B, c, h, w).permute(0, 2, 3, 1).contiguous().to(dev)
    protos = ops.mpm_protos(supn, mask.to(dev), ctr.to(dev), B, S, p)
    # protos_ref (adaptive_p) is [B,c,2p] ordered (fg0..,bg0..)
    got = protos.cpu().permute(0, 2, 1)
    # the soft-assignment logits are -|x - ctr|^2 ~ -c * 5 here: one fp32 ulp of them is c * 6e-7, and it moves the
    # softmax weights (hence the prototypes) by about as much -- the tolerance scales with c
    tol = 2e-5 * max(1.0, c / 128)
    assert torch.allclose(got, protos_ref, rtol=tol, atol=tol), (got - protos_ref).abs().max()
    pred, resp = ops.cosine_proto_max(qryn, protos, 20.0, want_resp=True)
    assert torch.allclose(pred.cpu(), pred_ref, rtol=0, atol=5e-5), (pred.cpu() - pred_ref).abs().max()
    if p == 3:      # the reference hard-codes "+3" for the fg response offset (pemp_stage1.py:221)
        # index work: EXACT wherever the winner's lead (inside its class and between the classes) exceeds twice the
        # value tolerance above; the pixels inside that margin are excluded and their fraction bounded
        from tests import util
        _, margin = util.response_reference(torch.cat([supn, qryn]), mask, ctr, B, S, p, 20.0, (h, w))
        util.assert_response_exact(resp, resp_ref, margin, margin=1e-4, max_masked=0.1 if h * w > 100 else 1.0, what=f"mpm {B}x{S} c{c} {h}x{w}")


@pytest.mark.parametrize("c,p", [(512, 3), (320, 3), (512, 5)])       # MFMA row stream / wave-per-pixel / MAXJ = 16
def test_soft_assignment_is_closer_to_float64_than_the_float32_formulation(hip_lib, dev, c, p):
    """The MPM assignment at the magnitudes trained features have (|x - c|^2 in the hundreds).  The reference's
    softmax(-sum (x - c)^2) in float32 carries the rounding of those sums (~1e-4 relative in the weights); the kernels
    evaluate the same softmax in its shift-invariant form (csrc/head_common.h) and must sit well inside that."""
    from pemp_amd import ops
    B, S, h, w = 1, 2, 13, 13
    n, J = h * w, 2 * p
    feat = _rand(B * S, h, w, c, seed=1, lo=-2.0, hi=2.0)
    ctr = _rand(c, J, seed=2, lo=-1.0, hi=1.0)
    m = (_rand(B * S, 1, h, w, seed=3) > 0.0).float()
    mask = torch.cat((m, 1 - m), dim=1)                                   # masks at feature resolution
    ws = {}
    ops.mpm_protos(feat.to(dev), mask.to(dev), ctr.to(dev), B, S, p, ws_cache=ws)
    A = ws[("mpm", B, S, h, w, c, p)][:B * S * J * n * 4].view(torch.float32).view(B * S, J, n).cpu().double()

    def assign(dt):
        x = feat.to(dt).reshape(B * S, n, c)
        D = -((x[:, :, None, :] - ctr.to(dt).t()[None, None]) ** 2).sum(-1)           # [BS, n, J]
        assert D.abs().mean() > 300
        P = torch.softmax(D.view(B * S, n, 2, p), dim=3).view(B * S, n, J).transpose(1, 2)
        return P * mask.to(dt).view(B * S, 2, 1, n).expand(-1, -1, p, -1).reshape(B * S, J, n)

    a64, a32 = assign(torch.float64), assign(torch.float32).double()
    big = a64 > 1e-3
    e_hip = ((A - a64).abs() / a64.clamp_min(1e-30))[big].max().item()
    e_ref = ((a32 - a64).abs() / a64.clamp_min(1e-30))[big].max().item()
    print(f"soft assignment c={c} p={p}: max relative error vs float64  hip {e_hip:.2e}   float32 reference form {e_ref:.2e}")
    assert e_hip < 5e-5 and e_hip < 0.5 * e_ref, (e_hip, e_ref)
    assert (A[a64 == 0] == 0).all()                                                     # masked-out pixels stay exactly zero


def test_masked_avg_pool_lowres_and_fullres(hip_lib, dev):
    from pemp_amd import ops
    B, S, c, h, w, H, W = 2, 2, 256, 7, 9, 50, 65
    sup = _rand(B * S, c, h, w, seed=1)
    m = (_rand(B * S, 1, H, W, seed=3) > 0.0).float()
    m[1] = 0                                    # empty foreground -> zero prototype
    mask = torch.cat((m, 1 - m), dim=1)
    supn = sup.permute(0, 2, 3, 1).contiguous().to(dev)
    # low-res (pemp_stage1.py:224-227)
    ml = F.interpolate(mask, (h, w), mode="nearest").view(B * S, 2, 1, h * w)
    f = sup.view(B * S, 1, c, h * w)
    ref = ((f * ml).sum(-1) / (ml.sum(-1) + 1e-5)).view(B, S, 2, c).mean(dim=1)
    got = ops.masked_avg_pool(supn, mask.to(dev), B, S, full_res=False).cpu()
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-6)
    # full-res (baseline.py:100-110)
    up = F.interpolate(sup, (H, W), mode="bilinear", align_corners=True)
    ref2 = torch.stack([(up * mask[:, g:g + 1]).sum(dim=(2, 3)) / (mask[:, g:g + 1].sum(dim=(2, 3)) + 1e-5)
                        for g in range(2)], dim=1).view(B, S, 2, c).mean(dim=1)
    got2 = ops.masked_avg_pool(supn, mask.to(dev), B, S, full_res=True).cpu()
    assert torch.allclose(got2, ref2, rtol=5e-5, atol=2e-6), (got2 - ref2).abs().max()
    assert (got2.view(B, 2, c)[0, 0].abs().max() > 0)


@pytest.mark.parametrize("h,w,Ho,Wo", [(13, 13, 97, 97), (51, 51, 333, 500), (51, 51, 500, 333), (5, 7, 5, 7), (4, 4, 1, 9)])
def test_upsample_and_eval_tail(hip_lib, dev, h, w, Ho, Wo):
    from pemp_amd import ops
    B = 2
    pred = _rand(B, 2, h, w, seed=1) * 5 + 15
    ref = F.interpolate(pred, (Ho, Wo), mode="bilinear", align_corners=True)
    got = ops.upsample_bilinear_ac(pred.to(dev), (Ho, Wo)).cpu()
    assert torch.allclose(got, ref, rtol=0, atol=2e-5), (got - ref).abs().max()
    tgt = (_rand(B, Ho, Wo, seed=2) > 0.3).long()
    tgt[0, : max(1, Ho // 7)] = 255
    am, stats, logits = ops.eval_tail(pred.to(dev), tgt.to(dev), want_logits=True)
    assert torch.equal(logits.cpu(), got)
    assert torch.equal(am.cpu().long(), got.argmax(dim=1))
    st = stats.cpu().numpy()
    for b in range(B):
        ce = F.cross_entropy(got[b:b + 1], tgt[b:b + 1], ignore_index=255, reduction="sum").item()
        nvalid = int((tgt[b] != 255).sum())
        assert st[b, 1] == nvalid
        assert abs(st[b, 0] - ce) <= 1e-5 * max(1.0, abs(ce))
        from tests.util import counts
        cn = counts(got[b].argmax(0).numpy(), tgt[b].numpy())
        assert (st[b, 2:].astype(np.int64) == cn.reshape(-1)).all()
    resp = (_rand(B, h, w, seed=3) * 3 + 3).to(torch.uint8)
    rref = F.interpolate(resp[:, None].float(), (Ho, Wo), mode="nearest")[:, 0].long()
    assert torch.equal(ops.upsample_nearest_u8_i64(resp.to(dev), (Ho, Wo)).cpu(), rref)


def test_cm_reduce(hip_lib, dev):
    from pemp_amd import ops
    N, C, Hm = 3, 64, 49
    mask = (_rand(N, 1, Hm, Hm, seed=1) > 0.5).float()
    for stride in (1, 2):
        mo = F.max_pool2d(mask, 3, stride, 1)
        h = mo.shape[-1]
        x = _rand(N, C, h, h, seed=2)
        masked = (x * mo).view(N, C, -1)
        got_m, stat = ops.cm_reduce(_nhwc(x).to(dev), mask[:, 0].to(dev), stride)
        assert torch.equal(got_m.cpu(), mo[:, 0])
        assert torch.allclose(stat.cpu()[:, 0], masked.mean(-1), rtol=1e-5, atol=1e-6)
        assert torch.equal(stat.cpu()[:, 1], masked.max(-1)[0])
    only = ops.cm_reduce(None, mask[:, 0].to(dev), 2)[0]
    assert torch.equal(only.cpu(), F.max_pool2d(mask, 3, 2, 1)[:, 0])


@pytest.mark.parametrize("N,C,h,w", [(3, 64, 25, 25), (2, 256, 13, 17), (5, 512, 7, 9), (2, 6, 11, 5)])   # C = 6: scalar kernels
def test_cm_statistics_and_their_adjoint(hip_lib, dev, N, C, h, w):
    """ResNetCM.comm statistics (backbones.py:208-216) and the gradient loss.backward() sends through them: masked mean
    and max per (image, channel); the max gradient goes to the FIRST maximal pixel (quantised inputs make ties)."""
    from pemp_amd import ops, train_ops as T
    mask = (_rand(N, 1, h, w, seed=1) > 0.3).float()
    x = ((_rand(N, C, h, w, seed=2) * 3).round() / 3).requires_grad_()
    mo = F.max_pool2d(mask, 3, 1, 1)
    masked = (x * mo).view(N, C, -1)
    mean, mx = masked.mean(-1), masked.max(-1)[0]
    xd = _nhwc(x.detach()).to(dev)
    got_m, stat = ops.cm_reduce(xd, mask[:, 0].to(dev), 1)
    assert torch.equal(got_m.cpu(), mo[:, 0])
    assert torch.allclose(stat.cpu()[:, 0], mean, rtol=1e-5, atol=1e-6) and torch.equal(stat.cpu()[:, 1], mx.detach())
    dstat = _rand(N, 2, C, seed=3)
    (mean * dstat[:, 0]).sum().backward(retain_graph=True)
    gmean = x.grad.clone()
    x.grad = None
    base = _rand(N, h, w, C, seed=4)
    dx = base.clone().to(dev)
    T.cm_bwd_add(xd, got_m, dstat.to(dev), dx)
    got = dx.cpu() - base
    # mean part everywhere + dmax at the first maximal pixel of x*mask (torch's CPU max picks that one too)
    first = masked.detach().argmax(-1)                                  # [N,C]
    want = _nhwc(gmean).clone()
    flat = want.view(N, h * w, C)
    mflat = mo.view(N, h * w)
    for n in range(N):
        flat[n, first[n], torch.arange(C)] += mflat[n, first[n]] * dstat[n, 1]
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (got - want).abs().max()
    if C % 4 == 0:      # training form: the forward records the winners, the backward is element-wise -- same bits
        m2, stat2, arg = ops.cm_reduce(xd, mask[:, 0].to(dev), 1, want_argmax=True)
        assert torch.equal(m2, got_m) and torch.equal(stat2, stat) and torch.equal(arg.cpu().long(), first)
        dx2 = base.clone().to(dev)
        T.cm_bwd_add(xd, got_m, dstat.to(dev), dx2, argmax=arg)
        assert torch.equal(dx2, dx)


def test_conv2d_group_equals_the_members_own_launches(hip_lib, dev):
    """pemp_conv2d_group_nhwc_f32: up to four independent convs in one launch, every member bit-identical to its own launch,
    for every group tile variant.  (a) the ASPPV2 branches of a small evaluation step: three dilated 3x3 convs with a folded
    BatchNorm (padding VALUE behind the activations) + the 1x1 branch, all reading the same x and writing slices of one
    concat buffer; (b) a stage's conv1 (ReLU, folded BN) beside its stride-2 downsample conv (no ReLU, different Cout);
    (c) a single member; and the argument checks."""
    from pemp_amd import ops, _lib
    N, HW, C = 2, 21, 128
    x = _rand(N, C, HW, HW, seed=11)
    flat = torch.empty(N * HW * HW + 8, C, device=dev)
    flat[:N * HW * HW].copy_(_nhwc(x).to(dev).view(-1, C))
    xin = flat[:N * HW * HW].view(N, HW, HW, C)
    qs, pvs = [], []
    for i, (k, dil) in enumerate(((1, 1), (3, 2), (3, 6), (3, 18))):
        w = _rand(C, C, k, k, seed=20 + i) * (1.0 / (C * k * k) ** 0.5)
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, _rand(C, seed=30 + i).to(dev), C, C, k, k, 1, dil if k == 3 else 0, dil, kpad, False, True)
        s = _rand(C, seed=40 + i, lo=0.5, hi=1.5)
        q, padv = ops.fold_input_affine(prm, s.to(dev), _rand(C, seed=50 + i).to(dev))
        flat[N * HW * HW + i].copy_(padv)
        qs.append(q)
        pvs.append(flat[N * HW * HW + i])
    ref = [ops.conv2d(xin, q, tile=23, pad_value=pv if q.kh > 1 else None) for q, pv in zip(qs, pvs)]
    order = [1, 2, 3, 0]
    for tile in ops.GROUP_TILES + (28,):
        if C % ops.TILE_VARIANTS[tile][1]:
            continue
        cat = torch.full((N, HW, HW, 4 * C), -3.0, device=dev)
        outs = [cat[..., i * C:(i + 1) * C] for i in range(4)]
        ops.conv2d_group([xin] * 4, [qs[i] for i in order], [outs[i] for i in order], pad_values=[pvs[i] for i in order], tile=tile)
        for i in range(4):
            assert torch.equal(outs[i], ref[i]), (tile, i)
    # (b) conv1 + downsample: different Cout, stride 2, ReLU on one member only, per-channel scale + shift on both
    Cin = 64
    xb = _nhwc(_rand(N, Cin, 33, 33, seed=60)).to(dev)
    mem = []
    for i, (co, relu) in enumerate(((64, True), (256, False))):
        w = _rand(co, Cin, 1, 1, seed=61 + i) * 0.1
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        mem.append(ops.ConvParams(packed, _rand(co, seed=63 + i, lo=0.5, hi=1.5).to(dev), _rand(co, seed=65 + i).to(dev), Cin, co, 1, 1, 2, 0, 1,
                                  kpad, False, relu))
    refb = [ops.conv2d(xb, p, tile=23) for p in mem]
    for tile in (23, 22, 25):
        got = ops.conv2d_group([xb, xb], mem, [torch.empty_like(r) for r in refb], tile=tile)
        assert all(torch.equal(g, r) for g, r in zip(got, refb)), tile
    assert bool((refb[0] >= 0).all()) and bool((refb[1] < 0).any())
    # (c) one member; autotuned call (tile = 0) leaves its pick in the cache and repeats bit for bit
    one = ops.conv2d_group([xb], [mem[1]], [torch.empty_like(refb[1])], tile=23)[0]
    assert torch.equal(one, refb[1])
    auto = ops.conv2d_group([xb, xb], mem, [torch.empty_like(r) for r in refb])
    assert all(torch.equal(g, r) for g, r in zip(auto, refb))
    # argument checks: five members, a shared output, mixed pad values, a tile the family does not have
    with pytest.raises(ValueError):
        ops.conv2d_group([xb] * 5, [mem[0]] * 5, [torch.empty_like(refb[0]) for _ in range(5)])
    same = torch.empty_like(refb[0])
    with pytest.raises(_lib.PempHipError, match="same output"):
        ops.conv2d_group([xb, xb], [mem[0], mem[0]], [same, same], tile=23)
    with pytest.raises(_lib.PempHipError, match="every member or for none"):
        ops.conv2d_group([xin, xin], [qs[1], qs[2]], [torch.empty_like(ref[1]), torch.empty_like(ref[2])], pad_values=[pvs[1], None], tile=23)
    with pytest.raises(_lib.PempHipError, match="21..28"):
        ops.conv2d_group([xb], [mem[0]], [torch.empty_like(refb[0])], tile=13)
    # members run beside each other in one grid: overlapping channel windows of one buffer (different base pointers), and an output
    # that is another member's input, are refused; disjoint windows of one buffer are what (a) does
    wide = torch.empty(N, HW, HW, 2 * C, device=dev)
    with pytest.raises(_lib.PempHipError, match="same output"):
        ops.conv2d_group([xin, xin], [qs[0], qs[0]], [wide[..., :C], wide[..., C // 2:C // 2 + C]], tile=23)
    ops.conv2d_group([xin, xin], [qs[0], qs[0]], [wide[..., :C], wide[..., C:]], tile=23)
    assert torch.equal(wide[..., :C], ref[0]) and torch.equal(wide[..., C:], ref[0])
    with pytest.raises(_lib.PempHipError, match="reads as its input"):
        ops.conv2d_group([xin, wide[..., :C]], [qs[0], qs[0]], [wide[..., :C], torch.empty_like(ref[0])], tile=23)
