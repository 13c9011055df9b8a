"""CELossDT on the device (weight map by exact EDT, weighted CE forward and backward) against the CPU
restatement of core/losses.py:17-43 (scipy's distance_transform_edt)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _targets():
    from pemp_amd import synth
    ts = [torch.from_numpy(synth.make_episode(s, out_hw=hw)["qry_mask"][0]) for s, hw in ((41, (97, 97)), (42, (97, 97)))]
    t = torch.stack(ts)
    t[0, :4, :9] = 255                       # ignored pixels count as background for the boundary
    big = torch.from_numpy(synth.make_episode(43, out_hw=(333, 500))["qry_mask"])
    empty = torch.zeros(1, 40, 57, dtype=torch.int64)      # no foreground: scipy's no-background case
    full = torch.ones(1, 9, 11, dtype=torch.int64)
    return [t, big, empty, full]


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_cedt_weight_matches_scipy(hip_lib, dev, idx):
    from oracle import ref_cpu
    from pemp_amd import ops
    t = _targets()[idx]
    ref = ref_cpu.cedt_weight(t, 5.0)
    got = ops.cedt_weight(t.to(dev), 5.0).cpu()
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6), (got - ref).abs().max()


def test_cedt_weight_and_loss_match_the_reference_class(hip_lib, dev):
    """Device weight map (boundary + exact EDT) and weighted CE vs what the reference's own CELossDT produced
    (tests/golden/cedt_reference.npz, make_golden.py --only cedt)."""
    from pemp_amd import ops
    from tests import util
    from tests.golden.cases import cedt_cases
    g = util.gold("cedt_reference")
    for n, (tgt, logits) in enumerate(cedt_cases()):
        w = ops.cedt_weight(tgt.to(dev), 5.0)
        assert torch.allclose(w.cpu(), torch.from_numpy(g[f"c{n}_weight"]), rtol=1e-6, atol=1e-6), n
        # the fused tail evaluates the loss from a low-resolution prediction; feed it the logits at identity scale
        _, stats, _ = ops.eval_tail(logits.to(dev).contiguous(), tgt.to(dev), weight=w)
        loss = (stats[:, 0].sum() / stats[:, 1].sum()).item()
        assert abs(loss - float(g[f"c{n}_loss"])) < 2e-6 * max(1.0, float(g[f"c{n}_loss"])), (n, loss)


def test_cedt_loss_and_gradient(hip_lib, dev):
    from oracle import ref_cpu
    from pemp_amd import ops, train_ops as T
    from tests.util import head_loss
    B, S, p, c, h, w, H = 2, 1, 3, 512, 13, 13, 97
    g = torch.Generator().manual_seed(3)
    feat = (torch.rand(B * S + B, h, w, c, generator=g) * 4 - 2).requires_grad_()
    m = (torch.rand(B, S, 1, H, H, generator=g) > 0.3).float()
    mask = torch.cat((m, 1 - m), dim=2)
    ctr = torch.rand(c, 2 * p, generator=g).requires_grad_()
    tgt = _targets()[0]
    wref = ref_cpu.cedt_weight(tgt, 5.0)
    loss, logits = head_loss(feat, mask, tgt, ctr, B, S, 1, p, 20.0, (97, 97), weight=wref)
    assert abs(loss.item() - ref_cpu.celoss_dt(logits.detach(), tgt, 5.0).item()) < 1e-6
    grads = torch.autograd.grad(loss, [feat, ctr])
    fd, md, td = feat.detach().to(dev), mask.reshape(B * S, 2, H, H).to(dev), tgt.to(dev)
    ws = {}
    pro = ops.mpm_protos(fd[:B * S], md, ctr.detach().to(dev), B, S, p, ws_cache=ws)
    pred = ops.cosine_proto_max(fd[B * S:], pro, 20.0)
    wmap = ops.cedt_weight(td, 5.0)
    _, stats, _ = ops.eval_tail(pred, td, ws_cache=ws, weight=wmap)
    assert abs((stats[:, 0].sum() / stats[:, 1].sum()).item() - loss.item()) < 1e-5
    dfeat = torch.empty_like(fd)
    dctr = T.head_bwd(fd[:B * S], fd[B * S:], md, ctr.detach().to(dev), ws[("mpm", B, S, h, w, c, p)], pro, pred, td, stats,
                      dfeat, B, S, p, 20.0, ws_cache=ws, weight=wmap)
    assert (dfeat.cpu() - grads[0]).abs().max().item() < 2e-4 * grads[0].abs().max().item()
    assert (dctr.cpu() - grads[1]).abs().max().item() < 5e-4 * grads[1].abs().max().item()


def test_losses_get_surface(hip_lib, dev):
    from pemp_amd.core import losses
    assert losses.get({"loss": "ce"}).kind == "ce" and losses.get({"loss": "cedt", "sigma": 5.0}).sigma == 5.0
    with pytest.raises(ValueError, match="Unsupported loss type"):
        losses.get({"loss": "dice"})
    logits = torch.randn(1, 2, 30, 40, device=dev)
    tgt = (torch.rand(1, 30, 40, device=dev) > 0.5).long()
    ref = torch.nn.functional.cross_entropy(logits.cpu(), tgt.cpu()).item()
    assert abs(losses.get({"loss": "ce"})(logits, tgt).item() - ref) < 1e-5


def test_train_step_with_cedt_runs(hip_lib, dev):
    from tests import util
    from tests.test_train_gpu import _batch
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr = Stage1Trainer(net, device=dev, drop_rate=0.0, loss="cedt", sigma=5.0)
    sup, msk, qry, gt = _batch(dev)
    l_hip, _ = tr.forward_backward(sup, msk, qry, gt)
    g_hip = tr.eng.flat.grad.clone()
    l_t, _ = util.torch_head_step(tr, sup, msk, qry, gt)
    assert abs(l_hip.item() - l_t.item()) < 2e-5
    assert (g_hip - tr.eng.flat.grad).abs().max().item() < 2e-2 * g_hip.abs().max().item()
