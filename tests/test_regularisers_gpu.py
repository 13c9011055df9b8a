"""DropBlock2D / Dropout2d kernels (dropout.hip).  The arithmetic applied to a given draw is checked EXACTLY
against the layers' formulas (dropblock==0.3.0 as described in SURVEY.md §8 a14; nn.Dropout2d) by passing the
uniforms in; the Philox stream itself is checked for determinism, stream separation and rate."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bs", [4, 3, 1, 6])
def test_dropblock_matches_layer_formula(hip_lib, dev, bs):
    from pemp_amd import train_ops as T
    torch.manual_seed(bs)
    n, h, w, c, p = 3, 29, 31, 32, 0.3
    u = torch.rand((n, h, w), device=dev)
    mask, cnt = T.dropblock_mask(n, h, w, p, bs, T.RandomStream(1), dev, uniforms=u)
    seed = (u < p / bs ** 2).float()[:, None]
    bm = F.max_pool2d(seed, kernel_size=bs, stride=1, padding=bs // 2)
    if bs % 2 == 0:
        bm = bm[:, :, :-1, :-1]
    bm = 1 - bm.squeeze(1)
    assert torch.equal(mask, bm) and int(cnt.item()) == int(bm.sum().item())
    x = torch.randn((n, h, w, c), device=dev)
    ref = x * bm[..., None]
    ref = ref * bm.numel() / bm.sum()
    assert torch.equal(T.pixel_scale(x, mask, cnt), ref)
    x2 = torch.randn((n * h * w, 8), device=dev)                       # 2-D rows form (the ASPP global branch)
    assert torch.equal(T.pixel_scale(x2, mask, cnt), x2 * bm.reshape(-1, 1) * bm.numel() / bm.sum())


def test_dropout2d_matches_layer_formula(hip_lib, dev):
    from pemp_amd import train_ops as T
    n, c, p = 5, 64, 0.5
    u = torch.rand((n, c), device=dev)
    m = T.dropout2d_mask(n, c, p, T.RandomStream(1), dev, uniforms=u)
    assert torch.equal(m, (u < 1 - p).float() / (1 - p))
    x = torch.randn((n, 7, 9, c), device=dev)
    assert torch.equal(T.channel_scale(x, m), x * m.view(n, 1, 1, c))
    wide = torch.randn((n, 7, 9, 2 * c), device=dev)
    out = torch.zeros_like(wide)
    T.channel_scale(wide[..., c:], m, out=out[..., :c])                  # strided views (channel slices)
    assert torch.equal(out[..., :c], wide[..., c:] * m.view(n, 1, 1, c)) and not out[..., c:].any()


def test_philox_stream_properties(hip_lib, dev):
    from pemp_amd import train_ops as T
    rs = T.RandomStream(1234, dev)
    a = T.dropout2d_mask(64, 1024, 0.5, rs, dev)
    b = T.dropout2d_mask(64, 1024, 0.5, rs, dev)                          # next offset -> different numbers
    assert not torch.equal(a, b)
    rs2 = T.RandomStream(1234, dev)
    assert torch.equal(T.dropout2d_mask(64, 1024, 0.5, rs2, dev), a)       # same (seed, offset) -> same mask
    assert not torch.equal(T.dropout2d_mask(64, 1024, 0.5, T.RandomStream(1235, dev), dev), a)
    keep = (a > 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01 and set(a.unique().tolist()) == {0.0, 2.0}
    # DropBlock seed rate: P(drop) = 1 - (1 - gamma)^(window) away from the border
    mask, cnt = T.dropblock_mask(8, 51, 51, 0.1, 4, T.RandomStream(7, dev), dev)
    frac = 1 - cnt.item() / mask.numel()
    assert 0.06 < frac < 0.11 and int(mask.sum().item()) == cnt.item()
    # the device-side step counter moves the stream without touching kernel arguments (hipGraph replays)
    rs3 = T.RandomStream(1234, dev)
    rs3.begin_step()
    c1 = T.dropout2d_mask(64, 1024, 0.5, rs3, dev)
    rs3.begin_step()
    c2 = T.dropout2d_mask(64, 1024, 0.5, rs3, dev)
    assert not torch.equal(c1, c2) and not torch.equal(c1, a)
    g = torch.cuda.CUDAGraph()
    rs4 = T.RandomStream(99, dev)
    static = {}
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        rs4.begin_step()
        static["m"] = T.dropout2d_mask(64, 1024, 0.5, rs4, dev)
    g.replay()
    m1 = static["m"].clone()
    g.replay()
    assert not torch.equal(m1, static["m"])


def test_dropblock_fused_into_its_neighbours_is_the_same_arithmetic(hip_lib, dev):
    """Round 4: DropBlock2D's scaling inside the kernel in front of it -- the conv epilogue (pemp_conv2d_dropblock_nhwc_f32:
    purifier conv -> bias -> ReLU -> DropBlock, and the input-gradient convs whose result DropBlock's backward scales) and the
    BatchNorm apply (pemp_bn_apply_dropblock_f32: ASPPV2's BN -> DropBlock -> conv) -- equals the separate pass bit for bit, for
    every tile variant the fused form runs on, split-K variants included."""
    from pemp_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(3)
    n, hw, cin, cout = 2, 31, 64, 128
    x = torch.randn(n, hw, hw, cin, generator=g).to(dev)
    w = (torch.randn(cout, 9 * cin, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    p = ops.ConvParams(w, None, b, cin, cout, 3, 3, 1, 2, 2, 9 * cin, False, True)
    u = torch.rand((n, hw, hw), generator=g).to(dev)
    rec = T.dropblock_mask(n, hw, hw, 0.3, 4, T.RandomStream(1), dev, uniforms=u)
    assert 0 < int(rec[1].item()) < n * hw * hw
    for tile in (23, 22, 25, 21, 24, 26, 31, 32, 34, 35, 36):
        ref = T.pixel_scale(ops.conv2d(x, p, tile=tile), *rec)
        got = ops.conv2d(x, p, tile=tile, dropblock=rec)
        assert torch.equal(got, ref), tile
    auto = ops.conv2d(x, p, dropblock=rec, splitk=True)               # autotuned pick among the fused variants
    assert torch.allclose(auto, ref, rtol=1e-4, atol=1e-4)            # (a split-K pick differs from the unsplit ones in rounding)
    # a geometry outside the buffer-addressed kernels (Cin = 16): conv + the layer's own pass, same result
    x16 = torch.randn(n, hw, hw, 16, generator=g).to(dev)
    # BatchNorm apply + DropBlock
    z = torch.randn(n, hw, hw, 256, generator=g).to(dev)
    mean, invstd = z.view(-1, 256).mean(0), 1.0 / (z.view(-1, 256).var(0, unbiased=False) + 1e-5).sqrt()
    gamma, beta = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev)
    ref = T.pixel_scale(T.bn_apply(z, mean, invstd, gamma, beta, torch.empty_like(z), relu=False), *rec)
    got = T.bn_apply_dropblock(z, mean, invstd, gamma, beta, torch.empty_like(z), rec)
    assert torch.equal(got, ref)
    del x16


@pytest.mark.parametrize("seeds,H", [((31, 32), 97), ((31, 32, 33, 34), 401)])
def test_training_step_with_fused_dropblock_equals_the_unfused_step(hip_lib, dev, monkeypatch, seeds, H):
    """The whole stage-1 training step with DropBlock active and GIVEN draws: fused (default) against PEMP_FUSE_DROPBLOCK=0 --
    same loss, same gradients (bit for bit with the kernel variants pinned to one pick per layer).  Also at BASELINE.json
    configs[2]'s per-rank shape (4 episodes, 401 x 401: 20 808 feature rows) -- the deterministic A/B the statistical
    fits-a-batch test of test_train_gpu.py cannot be."""
    from pemp_amd import ops, synth, train_engine as te
    from pemp_amd.networks import pemp_stage1 as m
    from tests import util
    b = synth.make_batch(list(seeds), shot=1, height=H, width=H, out_hw=(H, H))
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    batch = (t("sup_img"), t("sup_mask"), t("qry_img"), t("qry_mask")[:, 0])
    h = w = (H + 7) // 8
    nimg = 2 * len(seeds)
    gen = torch.Generator().manual_seed(77)
    layers = {"encoder.purifier.2": (h, w), "encoder.purifier.5": (h, w), "encoder.purifier.6.aspp_0.1": (1, 1)}
    layers.update({f"encoder.purifier.6.aspp_{i}.1": (h, w) for i in range(1, 5)})
    draws = {k: torch.rand((nimg,) + hw, generator=gen).to(dev) for k, hw in layers.items()}
    monkeypatch.setattr(ops, "AUTOTUNE", False)                        # one fixed variant per layer on both sides
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(te, "FUSE_DROPBLOCK", fuse)
        net = m.ModelClass(None)
        net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
        tr = te.Stage1Trainer(net, device=dev, drop_rate=0.3, block_size=4)
        tr.eng.draws = draws
        loss, _ = tr.forward_backward(*batch)
        torch.cuda.synchronize()
        res.append((float(loss), tr.eng.flat.grad.clone()))
    (l0, g0), (l1, g1) = res
    assert l0 == l1
    rel = ((g0 - g1).norm() / g1.norm()).item()
    print(f"fused vs unfused DropBlock: loss {l0:.6f}, relative gradient difference {rel:.2e}")
    assert rel < 1e-6
