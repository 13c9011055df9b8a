"""Integer probes of the conv kernels (csrc/conv_dma2.hip and its fall-backs): inputs, weights and padding values are small
integers, so every partial sum is an integer below 2^24 and fp32 addition is exact IN ANY ORDER -- every tile variant, the
split-K ones included, must equal the float64 CPU convolution EXACTLY, on tiles that are computed whole and on tiles that are
split.  What this catches that a comparison between variants on random data does not: a wrong tap displacement, a wrong
padding decision or a piece of K visited twice / not at all shows up as an exact integer error in every output it touches,
whichever variant it is in (round 4: one instantiation read its activations `tap` pixels off, with correct weights; found by
the variant comparison on one geometry only)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TILE_N = {21: 128, 22: 64, 23: 64, 24: 128, 25: 64, 26: 128, 27: 256, 28: 64, 29: 64, 31: 128, 32: 64, 34: 128, 35: 64, 36: 128, 37: 256,
          # the pointer-addressed LDS-DMA kernels (conv_dma.hip) and the first-generation kernel (conv_igemm.hip): the fall-backs
          11: 128, 12: 64, 13: 64, 14: 128, 15: 64, 16: 128, 17: 256, 1: 128, 2: 64, 3: 64}

# (N, H, W, Cin, dilation): 5202 rows = 82 tiles of 128 x 128 (every tile split); 20 808 rows = whole AND split tiles in one launch
# (326 tiles on 256 CUs); a non-square map under another dilation; a small map with dilation 1
GEOMS = [(2, 51, 51, 64, 6), (8, 51, 51, 64, 6), (3, 37, 45, 96, 2), (2, 23, 29, 64, 1)]


def _problem(N, H, W, Cin, d, Cout=256):
    m = torch.arange(N * H * W, dtype=torch.int64)
    c = torch.arange(Cin, dtype=torch.int64)
    x = ((m[:, None] * 7 + c[None, :] * 13) % 1021).view(N, H, W, Cin).double()            # NHWC integers in [0, 1021)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64)
    n = torch.arange(Cout)
    for t in range(9):                                   # tap t reads channel 5 t + 1 (+ 32 for odd output channels): weight 1..3
        w[n, (5 * t + 1 + 32 * (n % 2)) % Cin, t // 3, t % 3] = ((n + t) % 3 + 1).double()
    pv = (500 + 3 * c).double()                                                             # per-channel padding value
    return x, w, pv


def _reference(x, w, pv, d):
    """float64 conv2d of the NHWC integers with `pv` (or zero) outside the image -> NHWC."""
    N, H, W, Cin = x.shape
    xp = torch.empty(N, Cin, H + 2 * d, W + 2 * d, dtype=torch.float64)
    xp[:] = 0.0 if pv is None else pv.view(1, Cin, 1, 1)
    xp[:, :, d:d + H, d:d + W] = x.permute(0, 3, 1, 2)
    return F.conv2d(xp, w, None, 1, 0, d).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(str(v) for v in g))
@pytest.mark.parametrize("padv", [False, True], ids=["zero-padding", "padding-value"])
def test_every_variant_is_exact_on_integer_probes(hip_lib, dev, geom, padv):
    from pemp_amd import ops
    N, H, W, Cin, d = geom
    x, w, pv = _problem(N, H, W, Cin, d)
    ref = _reference(x, w, pv if padv else None, d)
    assert ref.abs().max().item() < 2 ** 24 and torch.equal(ref, ref.round())
    M = N * H * W
    buf = torch.empty(M + 4, Cin, device=dev)               # the padding vector sits behind the activations (PurifierEngine does the same)
    buf[:M] = x.view(M, Cin).float().to(dev)
    buf[M:] = pv.float().to(dev)
    xd, pvd = buf[:M].view(N, H, W, Cin), buf[M]
    packed, kpad = ops.pack_conv_weight(w.float().to(dev))
    prm = ops.ConvParams(packed, None, None, Cin, 256, 3, 3, 1, d, d, kpad, False, False)
    want = ref.float().to(dev)
    for tile, need in TILE_N.items():
        y = ops.conv2d(xd, prm, pad_value=pvd if padv else None, tile=tile)
        bad = (y != want)
        assert not bool(bad.any()), (geom, padv, tile, int(bad.sum()), (y - want)[bad][:4].tolist())
        y2 = ops.conv2d(xd, prm, pad_value=pvd if padv else None, tile=tile)               # and again: the same, launch after launch
        assert torch.equal(y2, want), (geom, padv, tile, "second launch")


def test_the_probe_sees_a_displaced_tap(hip_lib, dev):
    """The probe's sensitivity, shown on the reference side: one tap read one pixel off changes the result (so the equality above
    is not vacuous)."""
    N, H, W, Cin, d = GEOMS[0]
    x, w, pv = _problem(N, H, W, Cin, d)
    ref = _reference(x, w, pv, d)
    shifted = torch.roll(x, 1, dims=2)
    w_one = torch.zeros_like(w)
    w_one[:, :, 1, 2] = w[:, :, 1, 2]
    off = ref - _reference(x, w_one, pv, d) + _reference(shifted, w_one, pv, d)
    assert (off != ref).float().mean().item() > 0.5


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, Cout, k, pad, dilation (stride 1): training shapes of the buffer-addressed weight-gradient kernels
    (8, 51, 51, 128, 128, 3, 2, 2), (8, 51, 51, 128, 256, 1, 0, 1), (2, 101, 101, 64, 64, 3, 1, 1), (3, 37, 45, 128, 128, 3, 6, 6)],
    ids=lambda c: "x".join(str(v) for v in c))
def test_every_weight_gradient_variant_is_exact_on_integer_probes(hip_lib, dev, case):
    """The same idea for dW = sum over pixels of g^T x_col: small integers on both sides keep every partial sum below 2^24, so
    both kernel generations, both tile kinds and every block count (= every regrouping of the sum over pixels) must give the
    float64 result exactly."""
    from pemp_amd import ops, train_ops as T
    N, H, W, Cin, Cout, k, p, d = case
    M = N * H * W
    m = torch.arange(M, dtype=torch.int64)
    x = ((m[:, None] * 5 + torch.arange(Cin)[None, :] * 3) % 16).view(N, H, W, Cin).double()            # 0..15
    g = ((m[:, None] * 11 + torch.arange(Cout)[None, :] * 7) % 13 % 4).view(N, H, W, Cout).double()    # 0..3, H == Ho (same-size conv)
    assert 15 * 3 * M < 2 ** 24
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (Cout, Cin, k, k), g.permute(0, 3, 1, 2), 1, p, d)
    ref = ref.permute(0, 2, 3, 1).reshape(Cout, -1).float().to(dev)                                     # KRSC rows, as the kernels write them
    prm = ops.ConvParams(None, None, None, Cin, Cout, k, k, 1, p, d, k * k * Cin, False, False)
    xd, gd = x.float().to(dev), g.float().to(dev)
    picks = [(1, 0), (0, 0)] + [((kind, nb) if kind else nb, None) for kind in ((2, 3) if Cin % 128 == 0 and Cout % 128 == 0 else (0,))
                                for nb in T.WGRAD_BLOCK_CHOICES]
    for a, b in picks:
        dw = torch.full((Cout, k * k * Cin), float("nan"), device=dev)
        if b is None:
            T.conv_wgrad(xd, gd, prm, dw, blocks=a)
        else:
            T.conv_wgrad(xd, gd, prm, dw, variant=a, blocks=b)
        bad = dw != ref
        assert not bool(bad.any()), (case, a, b, int(bad.sum()), (dw - ref)[bad][:4].tolist())
