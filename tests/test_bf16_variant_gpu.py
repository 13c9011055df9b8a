"""The bf16-OPERAND variant of the stage-1 encoder (verdict r3 item 9: a side figure that tells what exact fp32 costs; never the
default path, never the headline).  Held here: the kernel computes what it says (bf16 operands, fp32 accumulation, one rounding
of the output), for every tile variant; the variant moves the prediction by what bf16 operands must move it and no more; the
fp32 path is untouched by its existence."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,HW,Cin,Cout,k,dil,res", [(2, 21, 64, 64, 1, 1, False), (2, 21, 128, 256, 3, 2, False),
                                                      (2, 33, 256, 128, 1, 1, True), (2, 51, 256, 256, 3, 6, False)])
def test_bf16_conv_is_bf16_operands_with_fp32_accumulation(hip_lib, dev, N, HW, Cin, Cout, k, dil, res):
    from pemp_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, HW, HW, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    r = torch.randn(N, HW, HW, Cout, generator=g).to(dev).to(torch.bfloat16) if res else None
    pad = dil if k == 3 else 0
    ref = F.conv2d(xb.float().permute(0, 3, 1, 2), wb.float(), b, 1, pad, dil).permute(0, 2, 3, 1)     # exact products of bf16 values
    if res:
        ref = ref + r.float()
    ref = F.relu(ref)
    packed = wb.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    p = ops.ConvParams(packed, None, b, Cin, Cout, k, k, 1, pad, dil, packed.shape[1], False, True)
    first = None
    for tile in ops.GROUP_TILES:
        if Cout % ops.TILE_VARIANTS[tile][1]:
            continue
        y = ops.conv2d(xb, p, residual=r, tile=tile)
        assert y.dtype == torch.bfloat16
        ulp = ref.abs().clamp_min(1e-3) * 2.0 ** -8               # half a bf16 ulp of the result, plus the fp32 sum's own noise
        assert bool(((y.float() - ref).abs() <= ulp + 1e-4).all()), tile
        if not res:
            y32 = ops.conv2d(xb, p, out=torch.empty(N, HW, HW, Cout, device=dev), tile=tile)
            assert (y32 - ref).abs().max().item() <= 3e-5 * (Cin * k * k / 64) ** 0.5, tile
        first = y if first is None else first
        assert torch.equal(y, first), tile                        # same K order in every variant
    # fp32 <-> bf16 copies: round to nearest even, exact way back
    v = torch.randn(1000, generator=g).to(dev)
    vb = ops.convert(v, torch.empty(1000, dtype=torch.bfloat16, device=dev))
    assert torch.equal(vb, v.to(torch.bfloat16))
    assert torch.equal(ops.convert(vb, torch.empty(1000, device=dev)), vb.float())


def test_bf16_variant_end_to_end_and_the_default_path_is_untouched(hip_lib, dev):
    from pemp_amd import synth
    from pemp_amd.networks import baseline as mb, pemp_stage1 as m
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()
    b = synth.make_batch([5678, 5679], shot=1, out_hw=(366, 500))
    t = lambda k_: torch.from_numpy(b[k_]).to(dev)
    sup, msk, qry = t("sup_img"), t("sup_mask"), t("qry_img")
    with torch.no_grad():
        p32 = net.lowres(sup, msk, qry)[0].clone()
        with net.precision("bf16"):
            p16 = net.lowres(sup, msk, qry)[0].clone()
            p16g = net.lowres_graphed(sup, msk, qry)[0].clone()
        again = net.lowres(sup, msk, qry)[0].clone()
    assert torch.equal(p32, again)                                 # the fp32 engine is a different object: bit for bit as before
    assert torch.equal(p16, p16g)
    d = (p32 - p16).abs()
    flips = float((p32.argmax(1) != p16.argmax(1)).float().mean())
    print(f"bf16 variant at 401 x 401: max |d pred| {d.max().item():.3f} of a range of 20, mean {d.mean().item():.4f}, "
          f"arg-max flips at feature resolution {flips:.4%}")
    assert 1e-4 < d.max().item() < 0.6 and d.mean().item() < 0.03 and flips < 0.02
    with pytest.raises(ValueError, match="stage-1 ResNet"):
        mb.Baseline(None, backbone="vgg16").precision("bf16")
    with pytest.raises(ValueError, match="precision"):
        net.precision("fp8")
