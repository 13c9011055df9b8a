"""The drop-in boundary in train() mode: ``model(sup, msk, qry, out_shape)`` returns logits with a
grad_fn, so the REFERENCE's Trainer body -- forward, ``loss_obj(...)``, ``loss.backward()``, ``clip_grad_norm_``,
``optimizer.step()`` with a stock torch optimizer (entry/pemp_stage1.py:57-65) -- runs unmodified on the HIP
training path.  Gradients are compared with the ones the reference produced (tests/golden/*_trainstep.npz) at the
tolerances of tests/test_train_gpu.py; the update is compared with the fused trainer's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu


def _batch(dev):
    from pemp_amd import synth
    b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]), b


def _check_grads(net, g, rtol_norm, rtol_t):
    params = dict(net.named_parameters())
    bad = []
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(name)]
        if ref < 0:
            assert p.grad is None or not p.requires_grad
            continue
        got = p.grad.norm().item()
        if abs(got - ref) > rtol_norm * ref + 1e-5:
            bad.append((str(name), got, float(ref)))
    assert not bad, bad[:10]
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        got = params[name].grad.cpu()
        ref = torch.from_numpy(g[key])
        got = (got if got.numel() <= 40000 else got.reshape(-1)[::37]).reshape(ref.shape)
        assert (got - ref).abs().max().item() <= rtol_t * max(ref.abs().max().item(), 1e-6) + 1e-7, name


def test_reference_trainer_body_runs_on_stage1(hip_lib, dev):
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    g = util.gold("stage1_rn50_trainstep")
    net = m.ModelClass(None, drop_rate=0.0).to(dev)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)     # core/solver.py:87-91
    sup, msk, qry, gt, _ = _batch(dev)
    # ---- the reference's train_step, verbatim in structure (entry/pemp_stage1.py:57-65) ----
    opt.zero_grad()
    qry_pred = net(sup, msk, qry, gt.shape[-2:])
    assert qry_pred.requires_grad and tuple(qry_pred.shape) == (2, 2, 97, 97)
    loss = F.cross_entropy(qry_pred, gt, ignore_index=255)
    loss.backward()
    total = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.1)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert (qry_pred.detach().cpu()[:, :, ::7, ::7].numpy() - g["logits_s7"]).__abs__().max() < 5e-3
    # gradients BEFORE clipping are what the fixture holds: undo the clip factor for the comparison
    coef = min(1.0, 1.1 / (total.item() + 1e-6))
    for p in net.parameters():
        if p.grad is not None:
            p.grad.div_(coef)
    util.check_gradients(g, util.gold("stage1_rn50_trainstep_f64"), dict(net.named_parameters()), "bridge stage1")
    for p in net.parameters():
        if p.grad is not None:
            p.grad.mul_(coef)
    opt.step()
    # ---- the same step on the fused trainer (one clip+SGD kernel) gives the same weights ----
    twin = m.ModelClass(None, drop_rate=0.0)
    twin.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr = Stage1Trainer(twin, device=dev, drop_rate=0.0)
    tr.train_step(sup, msk, qry, gt)
    for (k, a), (_, b) in zip(net.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-5, atol=2e-7), k
    for (k, a), (_, b) in zip(net.named_buffers(), twin.named_buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-7), k
    # ---- second step after zero_grad(set_to_none=True): p.grad views are re-attached; accumulation adds ----
    opt.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in net.parameters())
    F.cross_entropy(net(sup, msk, qry, (97, 97)), gt, ignore_index=255).backward()
    g1 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    F.cross_entropy(net(sup, msk, qry, (97, 97)), gt, ignore_index=255).backward()       # no zero_grad: accumulates
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-4, atol=1e-7), k
    # eval after training uses the updated weights and running statistics
    net.eval()
    with torch.no_grad():
        out = net(sup, msk, qry, (97, 97))
    assert torch.isfinite(out).all() and not out.requires_grad


def test_reference_trainer_body_runs_on_baseline_and_stage2(hip_lib, dev):
    from pemp_amd.networks import baseline as mb, pemp_stage2 as m2
    from tests.golden.cases import stage2_train_prior
    sup, msk, qry, gt, b = _batch(dev)
    for backbone, tag in (("vgg16", "baseline_vgg16"), ("resnet50", "baseline_rn50")):
        g = util.gold(tag + "_trainstep")
        net = mb.Baseline(None, backbone=backbone).to(dev)
        net.load_state_dict(util.wgen_state_dict(tag))
        net.train()
        loss = F.cross_entropy(net(sup, msk, qry, (97, 97)), gt, ignore_index=255)
        loss.backward()
        assert abs(loss.item() - float(g["loss"])) < 2e-5
        util.check_gradients(g, util.gold(tag + "_trainstep_f64"), dict(net.named_parameters()), "bridge " + tag,
                             eps=3e-3 if backbone == "vgg16" else 8e-3)          # see test_baseline_train_step_matches_reference
    g = util.gold("stage2_rn50cm_trainstep")
    net = m2.ModelClass(1, 1, None, drop_rate2=0.0).to(dev)
    net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    net.train()
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"])).to(dev)
    loss = F.cross_entropy(net(sup, msk, qry, prior, (97, 97)), gt, ignore_index=255)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    util.check_gradients(g, util.gold("stage2_rn50cm_trainstep_f64"), dict(net.named_parameters()), "bridge stage2")
    # Adam from the solver mirror steps on the same parameter views (tr.opt=adam, core/solver.py:92-96)
    from pemp_amd.core import solver
    opt, _ = solver.get(net, dict(solver.train_ingredient.cfg, opt="adam", adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8))
    before = net.ctr.detach().clone()
    opt.step()
    assert not torch.equal(before, net.ctr.detach()) and torch.isfinite(net.ctr).all()


def test_stage1_vgg16_trains_through_the_bridge(hip_lib, dev):
    """Stage 1 on VGG-16 (no purifier; MPM head): loss and gradients vs the reference's
    (tests/golden/stage1_vgg16_trainstep.npz), then the loss decreases over a few plain-SGD steps."""
    from pemp_amd.networks import pemp_stage1 as m
    net = m.ModelClass(None, backbone="vgg16").to(dev)
    net.load_state_dict(util.wgen_state_dict("stage1_vgg16"))
    net.train()
    g = util.gold("stage1_vgg16_trainstep")
    sup, msk, qry, gt, _ = _batch(dev)
    loss = F.cross_entropy(net(sup, msk, qry, (97, 97)), gt, ignore_index=255)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    util.check_gradients(g, util.gold("stage1_vgg16_trainstep_f64"), dict(net.named_parameters()), "bridge stage1 vgg16")
    net.zero_grad()
    opt = torch.optim.SGD(net.parameters(), lr=2e-3, momentum=0.9, weight_decay=5e-4)
    sup, msk, qry, gt, _ = _batch(dev)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = F.cross_entropy(net(sup, msk, qry, (97, 97)), gt, ignore_index=255)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.1)
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0], losses
