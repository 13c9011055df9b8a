"""PANet ("PANet reuse of the MAP/cosine kernels"): the Baseline's forward plus the
prototype-alignment branch, on the same HIP kernels with the roles of support and query swapped.  Parity against
vectors the reference's own networks/panet.py produced (tests/golden/panet_*.npz, make_golden.py --only panet) and
against the CPU oracle; training step against the reference's gradients of ``loss + align_loss``."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_models_gpu import _compare

pytestmark = pytest.mark.gpu


def _net(dev, backbone, tag):
    from pemp_amd.networks import panet as m
    net = m.PANet(None, backbone=backbone)
    net.load_state_dict(util.wgen_state_dict(tag))
    return net.to(dev)


@pytest.mark.parametrize("backbone,keys,fixtures", [
    ("vgg16", "panet_vgg16", ["panet_vgg16_small", "panet_vgg16_small5", "panet_vgg16_full"]),
    ("resnet50", "panet_rn50", ["panet_rn50_small"])])
def test_panet_matches_reference_golden(hip_lib, dev, backbone, keys, fixtures):
    net = _net(dev, backbone, keys).eval()
    for fx in fixtures:
        g = util.gold(fx)
        shot, H = int(g["shot"]), int(g["H"])
        for e, seed in enumerate(g["seeds"]):
            hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
            t = util.episode_tensors(seed, shot, H, hw, dev)
            with torch.no_grad():
                out, aux = net(t["sup_img"], t["sup_mask"], t["qry_img"], hw)
            _compare(g, e, out, t["qry_mask"], net._last_feats, H)
            # the branch's masks come from an argmax of near-tied logits at a few pixels: 2e-3 relative on the loss
            assert abs(aux.item() - float(g[f"e{e}_align_loss"])) < 2e-3 * max(1.0, float(g[f"e{e}_align_loss"])), (fx, e)


def test_panet_batch_matches_cpu_oracle_and_evaluator_contract(hip_lib, dev):
    """Two 2-shot episodes in one batch vs oracle/ref_cpu.panet_forward; Evaluator.test_step -> (pred, loss, aux_loss)."""
    from oracle import ref_cpu
    from pemp_amd import synth
    from pemp_amd.entry import panet as entry
    net = _net(dev, "vgg16", "panet_vgg16").eval()
    b = synth.make_batch([41, 42], shot=2, height=97, width=97, out_hw=(97, 97))
    sup, msk, qry = (torch.from_numpy(b[k]) for k in ("sup_img", "sup_mask", "qry_img"))
    gt = torch.from_numpy(b["qry_mask"])
    with torch.no_grad():
        ref_out, ref_aux = ref_cpu.panet_forward(util.wgen_state_dict("panet_vgg16"), sup, msk, qry, (97, 97))
        out, aux = net(sup.to(dev), msk.to(dev), qry.to(dev), (97, 97))
    assert (out.cpu() - ref_out).abs().max().item() < 5e-3
    assert abs(aux.item() - ref_aux.item()) < 2e-3 * max(1.0, ref_aux.item())
    ev = entry.Evaluator(net, device=dev, use_graph=False)
    pred, loss, aux2 = ev.test_step((sup, msk, qry), gt)
    ref_loss = torch.nn.functional.cross_entropy(ref_out, gt[:, 0], ignore_index=255).item()
    assert pred.shape == (2, 97, 97) and abs(loss - ref_loss) < 2e-4
    # test_step reports the mean over episodes of the per-episode branch loss
    per_ep = [ref_cpu.panet_forward(util.wgen_state_dict("panet_vgg16"), sup[i:i + 1], msk[i:i + 1], qry[i:i + 1], (97, 97))[1].item()
              for i in range(2)]
    assert abs(aux2 - float(np.mean(per_ep))) < 2e-3 * max(1.0, float(np.mean(per_ep)))
    util.assert_argmax_exact(out, ref_out.argmax(1), margin=1e-2, max_masked=0.02, what="panet")   # |d logit| < 5e-3 above
    assert (pred == out.cpu().argmax(1).numpy()).all()


@pytest.mark.parametrize("backbone,tag", [("vgg16", "panet_vgg16"), ("resnet50", "panet_rn50")])
def test_panet_train_step_matches_reference(hip_lib, dev, backbone, tag):
    """entry/panet.py:103-110: (loss + 1.0 * align_loss).backward() -- the branch's gradient reaches the encoder through
    pemp_head_bwd_f32 on swapped operands."""
    from pemp_amd import synth
    from pemp_amd.train_baseline import PANetTrainer
    g = util.gold(tag + "_trainstep")
    net = _net(dev, backbone, tag)
    tr = PANetTrainer(net, device=dev, loss_coef=1.0)
    b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
    t = lambda a: torch.from_numpy(a).to(dev)
    loss, _ = tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert abs(tr.last_align_loss.item() - float(g["align_loss"])) < 2e-3 * max(1.0, float(g["align_loss"]))
    params = dict(net.named_parameters())
    bad = []
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        if ref < 0:
            continue
        got = params[str(name)].grad.norm().item()
        if abs(got - ref) > 1e-2 * ref + 1e-5:
            bad.append((str(name), got, float(ref)))
    assert not bad, bad[:10]
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        got = params[name].grad.cpu()
        ref = torch.from_numpy(g[key])
        got = (got if got.numel() <= 40000 else got.reshape(-1)[::37]).reshape(ref.shape)
        assert (got - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item() + 1e-7, name


def test_panet_reference_trainer_body_runs_through_the_autograd_bridge(hip_lib, dev):
    """model(...) in train() mode returns (logits, align_loss), both differentiable: the reference's train_step body
    (entry/panet.py:104-109) runs as written and yields the gradients of the fused trainer."""
    from pemp_amd import synth
    from pemp_amd.train_baseline import PANetTrainer
    b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
    t = lambda a: torch.from_numpy(a).to(dev)
    ins = (t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]))
    gt = t(b["qry_mask"][:, 0])
    net = _net(dev, "vgg16", "panet_vgg16").train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()
    qry_pred, aux_loss = net(*ins, (97, 97))
    loss = torch.nn.functional.cross_entropy(qry_pred, gt, ignore_index=255)
    (loss + aux_loss * 0.5).backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    opt.step()
    ref_net = _net(dev, "vgg16", "panet_vgg16")
    tr = PANetTrainer(ref_net, device=dev, loss_coef=0.5)
    l2, _ = tr.forward_backward(*ins, gt)
    assert abs(loss.item() - l2.item()) < 1e-6 and abs(aux_loss.item() - tr.last_align_loss.item()) < 1e-6
    for k, p in ref_net.named_parameters():      # torch's CE backward vs the fused one: a few ulps apart
        assert (grads[k] - p.grad).abs().max().item() <= 1e-4 * p.grad.abs().max().item() + 1e-9, k


def _check_grad_fixture(net, g, rtol_norm=1e-2, rtol_t=1.5e-2):
    params = dict(net.named_parameters())
    bad = []
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        if ref < 0:
            continue
        got = params[str(name)].grad.norm().item()
        if abs(got - ref) > rtol_norm * ref + 1e-5:
            bad.append((str(name), got, float(ref)))
    assert not bad, bad[:10]
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        got = params[name].grad.cpu()
        ref = torch.from_numpy(g[key])
        got = (got if got.numel() <= 40000 else got.reshape(-1)[::37]).reshape(ref.shape)
        assert (got - ref).abs().max().item() <= rtol_t * ref.abs().max().item() + 1e-7, name


def test_five_shot_train_steps_match_reference_gradients(hip_lib, dev):
    """5-shot gradients from the reference: stage 1 (mean over shots inside the MPM, B = 2) and PANet (the alignment
    branch expanded over the S support images; the reference supports S > 1 only with one episode per batch)."""
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage1 as m1
    from pemp_amd.train_baseline import PANetTrainer
    from pemp_amd.train_engine import Stage1Trainer
    t = lambda a: torch.from_numpy(a).to(dev)
    b = synth.make_batch([41, 42], shot=5, height=97, width=97, out_hw=(97, 97))
    g = util.gold("stage1_rn50_trainstep5")
    net = m1.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr = Stage1Trainer(net, device=dev, drop_rate=0.0)
    loss, _ = tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    _check_grad_fixture(net, g)
    b = synth.make_batch([41], shot=5, height=97, width=97, out_hw=(97, 97))
    g = util.gold("panet_vgg16_trainstep5")
    net = _net(dev, "vgg16", "panet_vgg16")
    tr = PANetTrainer(net, device=dev, loss_coef=1.0)
    loss, _ = tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert abs(tr.last_align_loss.item() - float(g["align_loss"])) < 2e-3 * max(1.0, float(g["align_loss"]))
    _check_grad_fixture(net, g)
    # stage 2, 5-shot: the communication modules' episode means run over S + Q = 6 images
    from pemp_amd.networks import pemp_stage2 as m2
    from pemp_amd.train_stage2 import Stage2Trainer
    from tests.golden.cases import stage2_train_prior
    b = synth.make_batch([41, 42], shot=5, height=97, width=97, out_hw=(97, 97))
    g = util.gold("stage2_rn50cm_trainstep5")
    net = m2.ModelClass(5, 1, None)
    net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    tr = Stage2Trainer(None, net, device=dev, drop_rate2=0.0)
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"])).to(dev)
    loss, _ = tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]), prior)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    params = dict(net.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(name)]
        if ref < 0:
            assert not p.requires_grad, name
            continue
        # linear*.bias feeds a pre-BatchNorm activation: its true gradient is zero, both sides hold rounding noise
        assert abs(p.grad.norm().item() - ref) <= 1e-2 * ref + 2e-5, (str(name), p.grad.norm().item(), float(ref))
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        got = params[name].grad.cpu()
        ref = torch.from_numpy(g[key])
        got = (got if got.numel() <= 40000 else got.reshape(-1)[::37]).reshape(ref.shape)
        assert (got - ref).abs().max().item() <= 1.5e-2 * max(ref.abs().max().item(), 1e-6) + 2e-7, name
