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
