"""One training step of PEMP stage 1 on the HIP path against (a) the gradients the reference itself
produced (tests/golden/stage1_rn50_trainstep.npz: model.train(), batch-stat BN, DropBlock off) and
(b) torch SGD semantics for the update.

Tolerances: loss 2e-5; gradients are held to an fp64-RELATIVE bound (tests/util.py::check_gradients): with g64 the same
step evaluated in double precision (tests/golden/*_trainstep*_f64.npz, oracle under autograd) and ref32 the reference's
own fp32 gradients, every sampled tensor must satisfy  |hip - g64|_2 <= 3 |ref32 - g64|_2 + 3e-3 |g64|_2  and
|hip - g64|_oo <= 3 |ref32 - g64|_oo + 3e-3 |g64|_oo, every parameter's gradient norm  |n - n64| <= 2 |n_ref32 - n64| + 3e-3 n64
(also for VGG-16: a single arg-max switch in its stride-1 3x3 max pool -- two window elements 1e-6 apart -- moves the L2 error
of every gradient below it from 3e-6 to 1e-3, scratch/vgg_layerwise.py).  Measured: the reference's own fp32 gradients sit 1e-5 .. 6e-3 of max|g| from
fp64 (back-propagation through 50 batch-statistics BatchNorms, ReLU / max switches), the HIP path 1e-5 .. 1.2e-2.
Run at the fixture size (2 episodes, 97x97) and at the shape BASELINE.json configs[2] trains at (4 episodes,
401x401: 128x128 wgrad tiles, split-M reduce at M = 20.8k, ...).  Every kernel is checked separately at 1e-5..1e-4
in test_train_ops_gpu.py."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _trainer(dev, **kw):
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    return Stage1Trainer(net, device=dev, drop_rate=0.0, **kw), net


def _batch(dev, seeds=(31, 32), H=97, shot=1):
    from pemp_amd import synth
    b = synth.make_batch(list(seeds), shot=shot, height=H, width=H, out_hw=(H, H))
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])


def _fixture_batch(g, dev):
    seeds = [int(v) for v in g["seeds"]] if "seeds" in g.files else [31, 32]
    H = int(g["H"]) if "H" in g.files else 97
    shot = int(g["shot"]) if "shot" in g.files else 1
    return _batch(dev, seeds, H, shot), seeds, H, shot


_check_gradients = util.check_gradients


@pytest.mark.parametrize("head,fixture", [("hip", "stage1_rn50_trainstep"), ("torch", "stage1_rn50_trainstep"),
                                          ("hip", "stage1_rn50_trainstep_full")])
def test_train_step_gradients_match_reference(hip_lib, dev, head, fixture):
    """head="hip": the whole step on libpemp_hip.so; head="torch": encoder on HIP, head by autograd (cross-check).
    ``_full``: BASELINE.json configs[2]'s per-rank step -- 4 episodes, 401x401 -- against the reference's gradients."""
    from pemp_amd import ops
    g = util.gold(fixture)
    g64 = util.gold(fixture + "_f64")
    tr, net = _trainer(dev)
    (sup, msk, qry, gt), _, H, _ = _fixture_batch(g, dev)
    if head == "hip":
        loss, pred = tr.forward_backward(sup, msk, qry, gt)
        logits = ops.upsample_bilinear_ac(pred, (H, H))
    else:
        loss, logits = util.torch_head_step(tr, sup, msk, qry, gt)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert (logits.cpu()[:, :, ::7, ::7].numpy() - g["logits_s7"]).__abs__().max() < 1e-3
    params = dict(net.named_parameters())
    _check_gradients(g, g64, params, fixture)
    sd = net.state_dict()
    for key in [k for k in g.files if k.startswith("buf__")]:
        name = key[len("buf__"):]
        ref = torch.from_numpy(g[key])
        assert torch.allclose(sd[name].cpu().to(ref.dtype), ref, rtol=1e-4, atol=1e-5), name


@pytest.mark.parametrize("policy", range(8))
def test_train_step_under_every_admissible_pick(hip_lib, dev, monkeypatch, policy):
    """The kernel picks the autotuners may make change the rounding of a training step (split-K variants regroup the K sum, the
    weight-gradient kernels split the pixel rows over 512 .. 1536 blocks on 64 x 64 or 128 x 128 tiles).  The fixture tests run
    one pinned variant per layer (conftest.py::pinned_picks); here the SAME one-step bound (loss 2e-5, gradients by
    util.check_gradients against float64) is held under eight forced pick policies that between them put every split-K tile
    (31, 32, 34 .. 37), every plain tile family and every (tile kind, block count) of the weight gradient on every layer that
    admits it -- ops.PICK_HOOK decides instead of the timer, at any problem size."""
    from pemp_amd import ops
    fixture = "stage1_rn50_trainstep"
    g, g64 = util.gold(fixture), util.gold(fixture + "_f64")
    seen = {"conv": set(), "wgrad": set()}

    def hook(kind, cands, key):
        if kind == "conv":
            want = ops.SPLITK_TILES[policy % len(ops.SPLITK_TILES)]
            pick = want if want in cands else cands[(policy * 3 + 1) % len(cands)]
        else:
            pick = cands[policy % len(cands)]
        seen[kind].add(pick)
        return pick

    monkeypatch.setattr(ops, "PICK_HOOK", hook)
    tr, net = _trainer(dev)
    (sup, msk, qry, gt), _, H, _ = _fixture_batch(g, dev)
    loss, _ = tr.forward_backward(sup, msk, qry, gt)
    torch.cuda.synchronize()
    assert any(t > 30 for t in seen["conv"]) and seen["wgrad"], seen          # the policy really reached the kernels
    print(f"  policy {policy}: conv tiles {sorted(seen['conv'])}, weight-gradient picks {sorted(seen['wgrad'], key=str)}, loss {loss.item():.7f}")
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    _check_gradients(g, g64, dict(net.named_parameters()), fixture)


def test_optimizer_step_matches_torch_sgd(hip_lib, dev):
    """Same gradients -> fused clip+SGD on the flat buffer == clip_grad_norm_ + torch.optim.SGD."""
    tr, net = _trainer(dev)
    sup, msk, qry, gt = _batch(dev)
    tr.forward_backward(sup, msk, qry, gt)
    plist = [p for p in net.parameters() if p.requires_grad]
    ref = [torch.nn.Parameter(p.detach().clone().contiguous()) for p in plist]
    for r, p in zip(ref, plist):
        r.grad = p.grad.detach().clone().contiguous()
    opt = torch.optim.SGD(ref, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    total = torch.nn.utils.clip_grad_norm_(ref, 1.1)
    opt.step()
    tr.optimizer_step()
    assert abs(tr.last_grad_norm.item() - total.item()) <= 1e-5 * total.item()
    for r, p in zip(ref, plist):
        assert torch.allclose(p.detach(), r.detach(), rtol=1e-6, atol=1e-7)
    # state_dict still has the reference layout and loads into a fresh eval model
    from pemp_amd.networks import pemp_stage1 as m
    fresh = m.ModelClass(None)
    fresh.load_state_dict({k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()})


def _adam_f64_with_bound(w, g, m0, v0, coef, hp, step):
    """ATen's Adam update (torch/optim/adam.py: _single_tensor_adam == _multi_tensor_adam arithmetic) evaluated in float64
    on float32 inputs, with the float-narrowed scalars ATen uses, and a FORWARD ERROR BOUND per element for any float32
    evaluation of the same expression (u = 2^-24, one rounding per operation, first order):
        d  = coef g + wd w                         dd = u (2 |coef g| + |d|)
        m  = m0 + (1 - b1)(d - m0)                 dm = (1 - b1) dd + u ((1 - b1) |d - m0| + |m|)
        v  = b2 v0 + (1 - b2) d^2                  dv = (1 - b2)(2 |d| + dd) dd + u (b2 v0 + 2 (1 - b2) d^2 + v) + 2^-148
        q  = sqrt(v) / sqrt(bias2) + eps           dq = (sqrt(v + dv) - sqrt(max(v - dv, 0))) / sqrt(bias2) + 3 u q
        p  = w - s m / q                           dp = s (dm / q + (|m| + dm) dq / q^2) + 3 u |s m / q| + u |p|
    (dd can exceed |d| where the two terms of d cancel completely: the dd^2 and dm dq terms are kept for that.)
    For an element whose clipped gradient and weight-decay term cancel (|d| <~ eps) the step is lr d / (|d| + eps), slope
    lr / eps = 2e5: dd is a few ulp of |coef g|, not of |d|, and the bound carries exactly that amplification -- it is
    the reason a fixed rtol / atol cannot hold for every element of a 12 M-element gradient."""
    b1, b2 = hp["betas"]
    f32 = lambda x: float(np.float32(x))
    bias1, bias2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    s, c = f32(hp["lr"] / bias1), f32(bias2 ** 0.5)
    omb1, b2f, omb2, eps, wd = f32(1.0 - b1), f32(b2), f32(1.0 - b2), f32(hp["eps"]), f32(hp["weight_decay"])
    u = 2.0 ** -24
    w, g, m0, v0 = (t.double() for t in (w, g, m0, v0))
    gc = coef * g
    d = gc + wd * w
    dd = u * (2 * gc.abs() + d.abs())
    m = m0 + omb1 * (d - m0)
    dm = omb1 * dd + u * (omb1 * (d - m0).abs() + m.abs())
    v = b2f * v0 + omb2 * d * d
    dv = omb2 * (2 * d.abs() + dd) * dd + u * (b2f * v0 + 2 * omb2 * d * d + v) + 2.0 ** -148
    q = v.sqrt() / c + eps
    dq = ((v + dv).sqrt() - (v - dv).clamp_min(0).sqrt()) / c + 3 * u * q
    upd = s * m / q
    p = w - upd
    dp = s * (dm / q + (m.abs() + dm) * dq / (q * q)) + 3 * u * upd.abs() + u * p.abs()
    return p, m, v, dp, dm, dv


def test_fused_adam_step_matches_torch_adam(hip_lib, dev):
    """tr.opt = adam (reference core/solver.py:92-96): three updates by the fused clip + Adam kernel on the flat buffers
    against (a) ATen's update evaluated in float64 from the same float32 parameters, gradients and moments, EVERY element
    inside the derived forward-error bound of `_adam_f64_with_bound` (x 4: first-order terms, ATen's scalar division by
    reciprocal), no allowance for outliers; (b) torch.optim.Adam itself on a copy of the flat buffer, fed the gradient clipped
    by the kernel's own coefficient, held to the same bound -- which pins the float64 model to torch's semantics; (c)
    clip_grad_norm_'s total norm within 1e-5 relative (float32 sum of per-tensor norms against the kernel's double sum).
    Rounds 3-4 counted elements outside rtol 2e-6 / atol 2e-7 and allowed 12, then 40; a fresh box saw 80 -- the count of
    near-cancelling elements is a property of the gradient, not a constant (and the reference side clipped with torch's
    norm, 1e-5 away from the kernel's: another 1e-5 |g| of dd)."""
    tr, net = _trainer(dev)
    plist = [p for p in net.parameters() if p.requires_grad]
    hp = dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    tr.attach_optimizer(torch.optim.Adam(plist, **hp))
    f = tr.eng.flat
    assert f.data.numel() == sum(p.numel() for p in plist)
    ref = torch.nn.Parameter(f.data.detach().clone())
    opt_ref = torch.optim.Adam([ref], **hp)
    worst = 0.0
    for step, seeds in enumerate(((31, 32), (33, 34), (35, 36)), start=1):
        sup, msk, qry, gt = _batch(dev, seeds)
        tr.forward_backward(sup, msk, qry, gt)
        w0, g = f.data.detach().clone(), f.grad.detach().clone()
        if step == 1:
            m0, v0 = torch.zeros_like(w0), torch.zeros_like(w0)
        else:
            m0, v0 = f.exp_avg.clone(), f.exp_avg_sq.clone()
        total = torch.nn.utils.get_total_norm([p.grad for p in plist])        # what clip_grad_norm_ computes and returns
        tr.optimizer_step()
        norm = np.float32(tr.last_grad_norm.item())
        assert abs(float(norm) - total.item()) <= 1e-5 * total.item()
        # the kernel's clip coefficient, in its own float32 arithmetic (train_ops.hip: adam_clip_kernel)
        coef = np.minimum(np.float32(tr.max_norm) / (norm * np.float32(1.0) + np.float32(1e-6)), np.float32(1.0))
        assert coef < 1.0 or total.item() <= tr.max_norm
        p64, m64, v64, dp, dm, dv = _adam_f64_with_bound(w0, g, m0, v0, float(coef), hp, step)
        for name, got, want, tol in (("param", f.data, p64, dp), ("exp_avg", f.exp_avg, m64, dm), ("exp_avg_sq", f.exp_avg_sq, v64, dv)):
            ratio = ((got.double() - want).abs() / tol.clamp_min(1e-300)).max().item()
            worst = max(worst, ratio)
            assert ratio <= 4.0, (step, name, ratio)
        # torch.optim.Adam from the same state, gradient clipped by the same coefficient
        with torch.no_grad():
            ref.copy_(w0)
            if step > 1:
                opt_ref.state[ref]["exp_avg"].copy_(m0)
                opt_ref.state[ref]["exp_avg_sq"].copy_(v0)
        ref.grad = g * float(coef)
        opt_ref.step()
        st = opt_ref.state[ref]
        assert int(st["step"]) == step
        for name, got, want, tol in (("param", ref.detach(), p64, dp), ("exp_avg", st["exp_avg"], m64, dm), ("exp_avg_sq", st["exp_avg_sq"], v64, dv)):
            ratio = ((got.double() - want).abs() / tol.clamp_min(1e-300)).max().item()
            assert ratio <= 4.0, (step, "torch " + name, ratio)
        # and the parameters the model sees are the flat buffer (reference-shaped views)
        assert all(p.data_ptr() >= f.data.data_ptr() and p.data_ptr() < f.data.data_ptr() + f.data.numel() * 4 for p in plist)
    print("adam: worst |hip - f64| / bound = %.3f" % worst)
    assert tr.eng.flat.adam_step == 3 and not tr.optimizer.state      # the torch object only carries the hyper-parameters


def _clean_loss(tr, attr, *batch):
    """The loss of ``batch`` with the trainer's regulariser (``tr.eng.<attr>``: drop_rate / drop_rate2) switched OFF, no update:
    the deterministic quantity a stochastic training run is supposed to lower."""
    rate = getattr(tr.eng, attr)
    setattr(tr.eng, attr, 0.0)
    try:
        return tr.forward_backward(*batch)[0].item()
    finally:
        setattr(tr.eng, attr, rate)


def test_two_steps_reduce_loss_and_dropblock_runs(hip_lib, dev):
    """Eight steps WITH DropBlock drawing (rate 0.1) on one batch lower that batch's loss measured without DropBlock -- the
    per-step losses themselves scatter with the draws (rounds 2-5 asserted min(losses[4:]) < losses[0] on them: an outcome of
    the process-wide seed, which the engine reads at construction)."""
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr = Stage1Trainer(net, device=dev, lr=2e-3)            # default drop_rate 0.1: DropBlock active
    assert tr.eng.drop_rate == 0.1
    sup, msk, qry, gt = _batch(dev)
    before = _clean_loss(tr, "drop_rate", sup, msk, qry, gt)
    losses = [tr.train_step(sup, msk, qry, gt).item() for _ in range(8)]
    after = _clean_loss(tr, "drop_rate", sup, msk, qry, gt)
    print(f"  loss without DropBlock {before:.4f} -> {after:.4f}; steps with DropBlock {[round(l, 4) for l in losses]}")
    assert all(np.isfinite(losses)) and len(set(losses)) == 8 and after < before, (before, after, losses)


@pytest.mark.parametrize("rate,steps,loss_bound", [(0.0, 120, 0.06), (0.1, 150, None)])
def test_training_fits_a_fixed_batch(hip_lib, dev, rate, steps, loss_bound):
    """The step as a whole does what a training step is for: repeated on one batch of four episodes (Wgen weights, SGD lr 2e-3,
    momentum 0.9, clip 1.1) the loss falls from 0.58 to 0.025 in 120 steps and the evaluation of those episodes reaches a
    foreground IoU of 0.98 (measured; bounds 0.06 / 0.9).  With DropBlock drawing (rate 0.1) the median loss of the last twenty of
    150 steps stays under 0.5 x the first (measured 0.17 .. 0.24 against 0.64; single steps spike up to 0.40)."""
    from pemp_amd import ops, synth
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr = Stage1Trainer(net, device=dev, lr=2e-3, drop_rate=rate)
    sup, msk, qry, gt = _batch(dev, seeds=(31, 32, 33, 34))
    losses = [tr.train_step(sup, msk, qry, gt).item() for _ in range(steps)]
    assert all(np.isfinite(losses))
    if loss_bound is None:
        # the trajectory is chaotic in its details (DropBlock draws, timing-based kernel picks change the rounding): single
        # steps spike (seen: 0.17 .. 0.40 among the last ten), the level does not
        assert float(np.median(losses[-20:])) < 0.5 * losses[0], (losses[0], losses[-20:])
        return
    assert losses[-1] < loss_bound, losses[::10]
    net.eval()
    with torch.no_grad():
        pred, _ = net.lowres(sup, msk, qry)
        _, stats, _ = ops.eval_tail(pred, gt)
    st = stats.cpu().numpy()                                  # per episode: CE sum, pixels, bg tp/fp/fn, fg tp/fp/fn
    iou_fg = st[:, 5] / (st[:, 5] + st[:, 6] + st[:, 7])
    assert (iou_fg > 0.9).all(), iou_fg


@pytest.mark.parametrize("backbone,tag", [("vgg16", "baseline_vgg16"), ("resnet50", "baseline_rn50")])
def test_baseline_train_step_matches_reference(hip_lib, dev, backbone, tag):
    """Baseline (VGG16: conv+bias+ReLU chain and max-pool backward; ResNet-50: BN trunk + projection) with
    the full-resolution MAP head differentiated through its adjoint, vs the reference's gradients."""
    from pemp_amd.networks import baseline as m
    from pemp_amd.train_baseline import BaselineTrainer
    g = util.gold(tag + "_trainstep")
    net = m.Baseline(None, backbone=backbone)
    net.load_state_dict(util.wgen_state_dict(tag))
    tr = BaselineTrainer(net, device=dev)
    sup, msk, qry, gt = _batch(dev)
    loss, _ = tr.forward_backward(sup, msk, qry, gt)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    params = dict(net.named_parameters())
    # ResNet-50 Baseline at the fixture size: 676 samples per channel in layer3, and the head's gradient is nearly constant per
    # channel there -- the last block's BatchNorm backward subtracts it and amplifies fp32 noise 20x per BN (scratch/rn50_layerwise.py:
    # the BN kernel is 6e-8 from fp64 on its own inputs; dz is 2e-3 / 1.7e-2 / 1.2e-2 off after the first / second / every later
    # BN).  ATen's CPU BatchNorm accumulates in double (acc_type<float, false>), which the reference's fp32 run benefits from.
    _check_gradients(g, util.gold(tag + "_trainstep_f64"), params, tag, eps=3e-3 if backbone == "vgg16" else 8e-3)
    # the reference's baseline Trainer does not clip; the update must equal plain torch SGD
    plist = [p for p in net.parameters() if p.requires_grad]
    ref_p = [torch.nn.Parameter(p.detach().clone().contiguous()) for p in plist]
    for r, p in zip(ref_p, plist):
        r.grad = p.grad.detach().clone().contiguous()
    opt = torch.optim.SGD(ref_p, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    opt.step()
    tr.optimizer_step()
    for r, p in zip(ref_p, plist):
        assert torch.allclose(p.detach(), r.detach(), rtol=1e-6, atol=1e-7)


def _stage2_trainer(dev, shot=1, **kw):
    from pemp_amd.networks import pemp_stage2 as m
    from pemp_amd.train_stage2 import Stage2Trainer
    net = m.ModelClass(shot, 1, None)
    net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    return Stage2Trainer(None, net, device=dev, **kw), net


@pytest.mark.parametrize("fixture", ["stage2_rn50cm_trainstep", "stage2_rn50cm_trainstep5_full"])
def test_stage2_train_step_matches_reference(hip_lib, dev, fixture):
    """Stage 2 (4-channel stem, communication modules as per-image conv bias with explicit backward through the
    Linear / episode mean / masked mean+max statistics, trainable block BNs, ASPP without BN) vs the gradients
    the reference produced (tests/golden/stage2_rn50cm_trainstep*.npz; Dropout2d off) and their fp64 evaluation.
    ``_full``: one 5-shot episode at 401x401 (BASELINE.json configs[3]: 6 images, CM statistics at 101 x 101).
    (linear*: with one episode per batch the communication module's output is constant over the batch and the
    following batch-statistics BN removes it -- exact gradient 0, fp32 rounding noise on both sides.)"""
    from pemp_amd import ops
    from tests.golden.cases import stage2_train_prior
    from pemp_amd import synth
    g = util.gold(fixture)
    g64 = util.gold(fixture + "_f64")
    (sup, msk, qry, gt), seeds, H, shot = _fixture_batch(g, dev)
    tr, net = _stage2_trainer(dev, drop_rate2=0.0, shot=shot)
    b = synth.make_batch(seeds, shot=shot, height=H, width=H, out_hw=(H, H))
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"])).to(dev)
    loss, pred = tr.forward_backward(sup, msk, qry, gt, prior)
    logits = ops.upsample_bilinear_ac(pred, (H, H))
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert (logits.cpu()[:, :, ::7, ::7].numpy() - g["logits_s7"]).__abs__().max() < 1e-3
    params = dict(net.named_parameters())
    _check_gradients(g, g64, params, fixture)
    sd = net.state_dict()
    for key in [k for k in g.files if k.startswith("buf__")]:
        name = key[len("buf__"):]
        ref = torch.from_numpy(g[key])
        assert torch.allclose(sd[name].cpu().to(ref.dtype), ref, rtol=1e-4, atol=1e-5), name


def test_stage2_train_step_with_dropout2d_active_matches_oracle(hip_lib, dev):
    """Stage 2 with its regulariser ON (Dropout2d(0.5) after both purifier convs and after every ASPP branch): the seven layers
    get the SAME uniform draws on both sides -- the engine through ``eng.draws``, the oracle through its restatement of
    ATen's feature dropout (oracle/ref_cpu.py: Dropout2d) -- and the loss (2e-5) and every gradient (the fp64-relative bound
    of the other train-step tests, with the oracle's own float32 evaluation as ref32) must agree.  The engine draws the four
    spatial branches' masks in one [N, 4 * 256] call; the draws are given per reference layer."""
    from oracle import ref_cpu
    from tests.golden.cases import stage2_train_prior
    from pemp_amd import synth
    p, seeds, H = 0.5, [31, 32], 97
    tr, net = _stage2_trainer(dev, drop_rate2=p)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    sup, msk, qry, gt = _batch(dev, seeds, H)
    b = synth.make_batch(seeds, shot=1, height=H, width=H, out_hw=(H, H))
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"]))
    B, S, Q = 2, 1, 1
    perm = torch.tensor([x for e in range(B) for x in ([e * S + s for s in range(S)] + [B * S + e * Q + q for q in range(Q)])])
    layers = ["encoder.purifier.2", "encoder.purifier.5"] + [f"encoder.purifier.6.aspp_{i}.2" for i in range(5)]
    gen = torch.Generator().manual_seed(78)
    draws = {k: torch.rand((B * (S + Q), 256), generator=gen) for k in layers}
    tr.eng.draws = {k: v.to(dev) for k, v in draws.items()}
    loss, _ = tr.forward_backward(sup, msk, qry, gt, prior.to(dev))
    torch.cuda.synchronize()
    hip = {k: v.grad.detach().cpu().clone() for k, v in net.named_parameters() if v.requires_grad and v.grad is not None}
    ins = (sup.cpu(), msk.cpu(), qry.cpu(), gt.cpu())
    out = {}
    for dt in (torch.float32, torch.float64):
        do = ref_cpu.Dropout2d(p, {k: v[perm] for k, v in draws.items()})
        out[dt] = ref_cpu.step_gradients(sd, *ins, model="stage2", qry_prior=prior, dtype=dt, dropout2d=do)
        assert do.used == set(layers)
    assert abs(loss.item() - out[torch.float32][0]) < 2e-5 and abs(loss.item() - out[torch.float64][0]) < 2e-5
    util.check_gradients_live(hip, out[torch.float32][1], out[torch.float64][1], "stage2 + Dropout2d")
    # the draws mattered
    tr0, _ = _stage2_trainer(dev, drop_rate2=0.0)
    loss0, _ = tr0.forward_backward(sup, msk, qry, gt, prior.to(dev))
    assert abs(loss0.item() - loss.item()) > 1e-4


def test_stage2_train_steps_with_stage1_prior_and_dropout(hip_lib, dev):
    """Whole train_step: frozen stage-1 prior on the HIP eval path, Dropout2d active, SGD without clipping."""
    from pemp_amd.networks import pemp_stage1 as m1, pemp_stage2 as m2
    from pemp_amd.train_stage2 import Stage2Trainer
    s1 = m1.ModelClass(None)
    s1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    s1.to(dev).eval()
    net = m2.ModelClass(1, 1, None)
    net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    tr = Stage2Trainer(s1, net, device=dev, lr=2e-3)
    assert tr.max_norm == 0.0 and tr.eng.drop_rate2 == 0.5
    sup, msk, qry, gt = _batch(dev)
    s1_before = {k: v.detach().clone() for k, v in s1.state_dict().items()}
    w_before = tr.eng.flat.data.clone()
    losses = [tr.train_step(sup, msk, qry, qry_msk=gt).item() for _ in range(8)]
    print(f"  steps with Dropout2d(0.5): {[round(l, 4) for l in losses]}")
    # What this test states is the step's PLUMBING: every loss finite, the Dropout2d draws change from step to step (eight
    # different losses on one batch), the stage-2 weights move, the stage-1 prior network -- weights AND BatchNorm buffers: it
    # runs in eval mode -- is untouched, the result loads into a fresh model.  Rounds 2-5 also asserted min(losses[4:]) <
    # losses[0]: on two episodes with half the channels dropped, lr 2e-3 and no gradient clipping that was a coin flip of the
    # process-wide seed the engine reads at construction (the batch's loss WITHOUT Dropout2d goes 0.555 -> 0.572 over these
    # eight steps).  That a stage-2 step descends is stated without the regulariser by
    # test_five_shot_training_steps_stage1_and_stage2, the Dropout2d step's arithmetic by
    # test_stage2_train_step_with_dropout2d_active_matches_oracle.
    assert all(np.isfinite(losses)) and len(set(losses)) == 8, losses
    assert not torch.equal(tr.eng.flat.data, w_before)
    assert all(torch.equal(v, s1_before[k]) for k, v in s1.state_dict().items())
    fresh = m2.ModelClass(1, 1, None)
    fresh.load_state_dict({k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()})


def test_training_loop_epochs_eval_and_checkpoints(hip_lib, dev, tmp_path):
    """start_training_loop (reference core/base_trainer.py:183-294): epochs of fused train steps, evaluation every epoch,
    ckpt.pth / bestckpt.pth written as plain state_dicts that load back into a fresh model; poly LR steps per iteration."""
    import logging
    from pemp_amd.core.base_trainer import TrainingLoop
    from pemp_amd.entry import pemp_stage1 as e
    from pemp_amd.entry.train_stage1 import Trainer, synthetic_batches
    cfg = e.ex.apply_updates({"split": 0, "g.model_dir": str(tmp_path), "tr.total_epochs": 2, "tr.lrp": "poly", "tr.lr": 2e-3,
                              "data.train_n": 6, "data.bs": 2, "data.test_n": 10, "data.height": 97, "data.width": 97,
                              "te.epochs": 1, "data.test_bs": 2})
    try:
        net = e.ModelClass(None)
        net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
        tr = Trainer(net, lr=cfg["tr"]["lr"], device=dev)
        loop = TrainingLoop(cfg, tr, e.Evaluator(net, device=dev), logging.getLogger("t"), run_id=7)
        assert loop.steps_per_epoch == 3
        val = e.SyntheticEpisodes(10, 5678, 1, 0, 97, 97)          # 10 episodes cover the 5 validation classes of the split
        hist = loop.start_training_loop(lambda ep: synthetic_batches(2, 1, 3, 100 + ep, 0, 97, 97), val, 20, 0)
    finally:
        for ing in e.INGREDIENTS + [e.ex]:
            ing._updates.clear()
            ing._cfg = None
    assert len(hist) == 2 and all(np.isfinite(h["train_loss"]) and 0 <= h["val_mIoU"] <= 1 for h in hist) and hist[0]["best"]
    assert loop.optimizer.param_groups[0]["lr"] < 2e-3                       # poly decays per iteration
    d = tmp_path / "pemp_stage1" / "7"
    assert sorted(p.name for p in d.iterdir()) == ["bestckpt.pth", "ckpt.pth"]
    fresh = e.ModelClass(None)
    fresh.load_weights(d / "ckpt.pth", logging.getLogger("t"))
    for (k, a), (_, b) in zip(fresh.state_dict().items(), net.state_dict().items()):
        assert torch.equal(a, b.cpu()), k


def test_five_shot_training_steps_stage1_and_stage2(hip_lib, dev):
    """5-shot episodes (6 images per episode; the MPM mean over shots, comm-module episode means over S + Q = 6):
    a few fused steps stay finite and reduce the loss on a fixed batch."""
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage1 as m1, pemp_stage2 as m2
    from pemp_amd.train_engine import Stage1Trainer
    from pemp_amd.train_stage2 import Stage2Trainer
    b = synth.make_batch([41, 42], shot=5, height=97, width=97, out_hw=(97, 97))
    t = lambda a: torch.from_numpy(a).to(dev)
    sup, msk, qry, gt = t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])
    assert sup.shape[1] == 5
    net1 = m1.ModelClass(None)
    net1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    tr1 = Stage1Trainer(net1, device=dev, lr=2e-3, drop_rate=0.0)
    l1 = [tr1.train_step(sup, msk, qry, gt).item() for _ in range(6)]
    assert all(np.isfinite(l1)) and min(l1[3:]) < l1[0], l1
    net2 = m2.ModelClass(5, 1, None)
    net2.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    tr2 = Stage2Trainer(net1.eval(), net2, device=dev, lr=2e-3, drop_rate2=0.0)
    l2 = [tr2.train_step(sup, msk, qry, gt).item() for _ in range(6)]
    assert all(np.isfinite(l2)) and min(l2[3:]) < l2[0], l2


def test_overlapped_gradient_buckets_on_one_rank_match_the_plain_step(hip_lib, dev, monkeypatch):
    """The bucketed all-reduce issued DURING backward (RCCL under the side stream, see GradBuckets) leaves the same
    weights as the plain step.  One rank: the collectives are identities, but the event / stream choreography, the
    bucket bookkeeping and finish() are the ones an 8-GPU job runs."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    except Exception as e:                                   # no RCCL in this environment: nothing to exercise
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        sup, msk, qry, gt = _batch(dev)
        weights, launched = [], []
        for force in (False, True):
            if force:
                monkeypatch.setenv("PEMP_FORCE_BUCKETS", "1")
            tr, net = _trainer(dev, lr=2e-3)
            bk = tr.eng.buckets
            assert len(bk.buckets) >= 3 and bk.buckets[0][1] == tr.eng.flat.n and bk.buckets[-1][0] == 0
            seen = []
            orig = bk._launch
            bk._launch = lambda lo, hi, orig=orig, seen=seen: (seen.append((lo, hi)), orig(lo, hi))[1]
            losses = [tr.train_step(sup, msk, qry, gt) for _ in range(3)]
            torch.cuda.synchronize()
            assert all(torch.isfinite(l) for l in losses)
            launched.append(seen)
            weights.append(tr.eng.flat.data.clone())
        assert launched[0] == [] and launched[1] == list(tr.eng.buckets.buckets) * 3     # every bucket once per step
        assert torch.equal(weights[0], weights[1])
    finally:
        dist.destroy_process_group()


def test_bench_collectives_run_on_rccl_with_one_rank(hip_lib, dev):
    """Every collective call pattern of the N > 1 bench lines and of the evaluation loop, on RCCL itself with one rank (the
    collectives are identities; what is checked is that RCCL on this platform takes these dtypes and call forms, that the
    results come back on the device, and that the device is usable after the group is destroyed, as bench.py's rank 0
    needs it): the round table's float64 all-reduce, float64 all-gather and MAX all-reduce of the timing scalars, broadcasts of
    the fp32 flat buffer and of an int64 BatchNorm counter, the tile picks' object broadcast, barrier."""
    import socket
    import torch.distributed as dist
    import bench
    from pemp_amd import ops
    from pemp_amd.entry.pemp_stage1 import DeviceRoundTable
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    except Exception as e:
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        dist.barrier()
        table = DeviceRoundTable(20, dev)
        table.reset()
        rows = torch.tensor([[5.0, 2.0, 1.0, 7.0, 3.0, 2.0, 4.0, 1.0]] * 3, dtype=torch.float64, device=dev)
        table.add(rows, torch.tensor([1, 4, 4], device=dev))
        before = table.pack.clone()
        assert before.dtype == torch.float64
        dist.all_reduce(table.pack, op=dist.ReduceOp.SUM)
        assert torch.equal(table.pack, before)
        got = [torch.zeros_like(before)]
        dist.all_gather(got, before)
        assert torch.equal(got[0], before)
        assert bench.gather_rank_ms(0.5, 10, 1, dev) == [50.0]
        tmax = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        assert float(tmax.item()) == 1.25
        flat = torch.randn(1 << 20, device=dev)
        keep = flat.clone()
        dist.broadcast(flat, 0)
        counter = torch.tensor(7, dtype=torch.int64, device=dev)
        dist.broadcast(counter, 0)
        assert torch.equal(flat, keep) and int(counter.item()) == 7
        box = [{"picks": {"(1, 2, 3)": 27}, "wgrad": [[2, 512]]}]
        dist.broadcast_object_list(box, src=0)
        assert box[0]["picks"]["(1, 2, 3)"] == 27
        assert bench.rccl_version() is not None
        assert bench.comm_object(1, [50.0])["rank_ms_per_step"]["max"] == 50.0
        dist.barrier()
    finally:
        dist.destroy_process_group()
    # the device after the group is gone (bench.py's rank 0 measures its roofline then)
    x = torch.randn(1, 13, 13, 64, device=dev)
    w = torch.randn(64, 64, 1, 1, device=dev)
    pk, kp = ops.pack_conv_weight(w)
    y = ops.conv2d(x, ops.ConvParams(pk, None, None, 64, 64, 1, 1, 1, 0, 1, kp, False, False))
    assert torch.isfinite(y).all()


def test_train_command_writes_a_fresh_run_and_test_command_loads_it(hip_lib, dev, tmp_path):
    """The command layer end to end (reference entry/pemp_stage1.py:68-113,116-167): ``train`` writes checkpoints into a
    FRESH run directory (never into an existing one), ``test with exp_id=<run>`` finds that checkpoint with the
    reference's lookup rules and evaluates THOSE weights; ``test`` without any checkpoint is an error, not an evaluation
    of random weights."""
    from pemp_amd.entry import pemp_stage1 as e
    common = ["split=0", f"g.model_dir={tmp_path}", "data.height=97", "data.width=97", "data.test_n=6", "te.epochs=1", "data.test_bs=2"]

    def run(*argv):
        try:
            return e.ex.run_commandline(["prog", *argv])
        finally:
            for ing in e.INGREDIENTS + [e.ex]:
                ing._updates.clear()
                ing._cfg = None

    with pytest.raises(FileNotFoundError):
        run("test", "with", *common)
    (tmp_path / "pemp_stage1" / "1").mkdir(parents=True)           # an older run exists: the new one must not reuse its id
    msg = run("train", "with", *common, "tr.total_epochs=1", "data.train_n=4", "data.bs=2", "exp_id=1")
    d = tmp_path / "pemp_stage1" / "2"
    assert sorted(p.name for p in d.iterdir()) == ["bestckpt.pth", "ckpt.pth"] and not list((tmp_path / "pemp_stage1" / "1").iterdir())
    assert "pemp_stage1/2" in msg.replace("\\", "/")
    out = run("test", "with", *common, "exp_id=2")
    assert out.startswith("Loss:") and "mIoU" in out
    # the evaluated weights are the checkpoint's: the same numbers as an evaluator fed from the file directly
    net = e.ModelClass(None)
    net.load_weights(d / "bestckpt.pth", __import__("logging").getLogger("t"))
    ev = e.Evaluator(net.to(dev).eval(), device=dev)
    loss, miou, biou = ev.start_eval_loop(e.SyntheticEpisodes(6, 5678, 1, 0, 97, 97), 20, 0, te_epochs=1, batch=2)
    assert out == f"Loss: {loss:.4f}, mIoU: {np.mean(miou) * 100:.2f}, bIoU: {np.mean(biou) * 100:.2f}"
    assert run("test", "with", *common, "ckpt=wgen").startswith("Loss:")


@pytest.mark.parametrize("model", ["stage1", "stage2", "baseline"])
def test_segmented_graph_step_equals_the_eager_step(hip_lib, dev, model):
    """The training step replayed from a CHAIN of hipGraphs (train_engine.SegmentedCapture: main-stream segments + side-stream
    weight-gradient segments that run beside the next main segment) leaves bit-identical weights, losses and BatchNorm buffers
    to the eager two-stream step -- regularisers off (their Philox stream advances identically, but keep the check about the
    graphs) -- over several replays with changing inputs."""
    from pemp_amd import synth
    from pemp_amd.networks import baseline as mb, pemp_stage1 as m1, pemp_stage2 as m2
    from pemp_amd.train_baseline import BaselineTrainer
    from pemp_amd.train_engine import Stage1Trainer
    from pemp_amd.train_stage2 import Stage2Trainer
    from tests.golden.cases import stage2_train_prior
    batches = []
    for s in range(5):
        b = synth.make_batch([31 + 2 * s, 32 + 2 * s], shot=1, height=97, width=97, out_hw=(97, 97))
        t = lambda a: torch.from_numpy(a).to(dev)
        batches.append((t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]), t(stage2_train_prior(b["qry_mask"]))))
    out = []
    for use_graph in (False, True):
        if model == "stage1":
            net = m1.ModelClass(None)
            net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
            tr = Stage1Trainer(net, device=dev, lr=2e-3, drop_rate=0.0, use_graph=use_graph)
        elif model == "stage2":
            net = m2.ModelClass(1, 1, None)
            net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
            tr = Stage2Trainer(None, net, device=dev, lr=2e-3, drop_rate2=0.0, use_graph=use_graph)
        else:
            net = mb.Baseline(None, backbone="vgg16")
            net.load_state_dict(util.wgen_state_dict("baseline_vgg16"))
            tr = BaselineTrainer(net, device=dev, lr=2e-3, use_graph=use_graph)
        losses = []
        for sup, msk, qry, gt, prior in batches:
            losses.append(tr.train_step(sup, msk, qry, gt, prior) if model == "stage2" else tr.train_step(sup, msk, qry, gt))
        torch.cuda.synchronize()
        if use_graph:
            cap = next(iter(tr._graphs.values()))["cap"]
            assert len(cap.main) >= 3 and sum(g is not None for g in cap.side) >= 3        # really a chain, really two streams
        out.append((torch.stack(losses).cpu(), tr.eng.flat.data.clone().cpu(),
                    {k: v.clone().cpu() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}))
    (l0, w0, b0), (l1, w1, b1) = out
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(w0, w1)
    assert all(torch.equal(b0[k], b1[k]) for k in b0)


@pytest.mark.autotuned
def test_full_size_training_steps_are_reproducible(hip_lib, dev):
    """BASELINE config 3 at its real shape (4 episodes of 401 x 401 per step, DropBlock on): two trainers started from the same
    weights take the same three steps bit for bit -- split-K fix-ups in arrival order, the two-stream overlap, the statistics
    partials and the weight-gradient splits all reduce in a fixed order (the kernel picks of the first trainer are reused by
    the second: same process, same cache)."""
    from pemp_amd import ops, synth
    from pemp_amd.networks import pemp_stage1 as m1
    from pemp_amd.train_engine import Stage1Trainer
    batches = []
    for s in range(3):
        b = synth.make_batch([700 + 4 * s + i for i in range(4)], shot=1, out_hw=(401, 401))
        t = lambda a: torch.from_numpy(a).to(dev)
        batches.append((t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])))
    out = []
    for _ in range(2):
        torch.manual_seed(1234)                      # the DropBlock stream is seeded from torch's seed at construction
        net = m1.ModelClass(None)
        net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
        tr = Stage1Trainer(net, device=dev, lr=2e-3)
        if not out:                                  # first trainer: one throw-away step fills the autotune caches
            warm = Stage1Trainer(m1.ModelClass(None), device=dev)
            warm.train_step(*batches[0])
        losses = [tr.train_step(*bt) for bt in batches]
        torch.cuda.synchronize()
        out.append((torch.stack(losses).cpu(), tr.eng.flat.data.clone().cpu()))
    assert any(t > 30 for t in ops._TILE_CACHE.values()), "no split-K variant was picked at the training shape"
    assert torch.isfinite(out[0][0]).all()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.autotuned
def test_full_size_graph_chain_equals_the_eager_step(hip_lib, dev):
    """The hipGraph chain against the eager step at BASELINE config 3's real shape (4 episodes of 401 x 401): the split-K
    variants (arrival counters reset by the kernels themselves, uncached workspace) and the per-shape weight-gradient picks are
    part of the recorded segments; weights after five steps (two eager warm-ups + three replays) are bit-identical."""
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage1 as m1
    from pemp_amd.train_engine import Stage1Trainer
    batches = []
    for s in range(5):
        b = synth.make_batch([900 + 4 * s + i for i in range(4)], shot=1, out_hw=(401, 401))
        t = lambda a: torch.from_numpy(a).to(dev)
        batches.append((t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])))
    warm = Stage1Trainer(m1.ModelClass(None), device=dev, drop_rate=0.0)
    warm.train_step(*batches[0])                     # fills the autotune caches: both trainers below replay the same picks
    out = []
    for use_graph in (False, True):
        net = m1.ModelClass(None)
        net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
        tr = Stage1Trainer(net, device=dev, lr=2e-3, drop_rate=0.0, use_graph=use_graph)
        losses = [tr.train_step(*bt) for bt in batches]
        torch.cuda.synchronize()
        if use_graph:
            cap = next(iter(tr._graphs.values()))["cap"]
            assert len(cap.main) >= 10 and sum(g is not None for g in cap.side) >= 10
        out.append((torch.stack(losses).cpu(), tr.eng.flat.data.clone().cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_five_step_trajectory_matches_the_reference(hip_lib, dev):
    """FIVE consecutive training steps (the loop of core/base_trainer.py:194-200 around entry/pemp_stage1.py:57-65): fused
    clip + momentum SGD on the flat buffers, BatchNorm running statistics, a different batch every step -- against the
    reference model stepped by torch.optim.SGD(lr 1e-3, momentum 0.9, weight decay 5e-4) + clip_grad_norm_(1.1) on the CPU
    (tests/golden/stage1_rn50_trajectory.npz; the oracle reproduces it to 2e-6 / 1e-6, test_cpu_suite.py).

    Bounds.  This trajectory amplifies rounding from step to step (the gradient norm jumps 15 -> 37.5 at step 4): the fixture
    therefore carries its own ENVELOPE -- the same reference code run from 8 copies of the initial weights perturbed by a
    relative 1e-7 N(0,1), about one float32 ulp (make_golden.py::gen_train_trajectory): per step the largest |d loss| and
    relative |d norm|, per tensor the largest relative L-inf distance of the final weights.  Step 4 of those replicas moves
    by up to 3.3e-2 in the norm and 4.7e-5 in the loss (steps 0-2: <= 6e-4 / 4e-6), so rounds 4-5's fixed 2e-2 / 1e-4 sat
    INSIDE what one ulp does to the reference itself.  Every step is now held to  C x its own envelope + the one-step floor
    (C = 3: the HIP path differs from ATen by the rounding of every accumulation, not by one ulp of the weights; floors: loss
    2e-6 -- what the oracle keeps against the fixture --, norm 2e-4, weights 2e-6, running statistics 1e-4 -- the one-step fixtures' tolerance); steps 0-2, where a wrong momentum /
    weight-decay / running-statistics update would show first, stay below 1e-3 in the norm and 1.4e-5 in the loss by that
    rule, and are additionally capped at those values.  The kernel picks are pinned (conftest.py::pinned_picks), so the
    arithmetic is the same on every box."""
    C_ENV, FLOOR_LOSS, FLOOR_NORM, FLOOR_W = 3.0, 2e-6, 2e-4, 2e-6
    g = util.gold("stage1_rn50_trajectory")
    tr, net = _trainer(dev, lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.1)
    bad = []
    for step in range(int(g["steps"])):
        sup, msk, qry, gt = _batch(dev, seeds=(31 + 2 * step, 32 + 2 * step))
        loss = tr.train_step(sup, msk, qry, gt).item()
        dl = abs(loss - float(g["losses"][step]))
        dn = abs(float(tr.last_grad_norm) - float(g["grad_norms"][step])) / float(g["grad_norms"][step])
        bl = C_ENV * float(g["env_loss"][step]) + FLOOR_LOSS
        bn = C_ENV * float(g["env_norm"][step]) + FLOOR_NORM
        if step <= 2:
            bl, bn = min(bl, 1.4e-5), min(bn, 1e-3)
        print(f"  step {step}: loss {loss:.6f} (reference {float(g['losses'][step]):.6f}; |d| {dl:.2e} <= {bl:.2e}), gradient norm "
              f"{float(tr.last_grad_norm):.4f} ({float(g['grad_norms'][step]):.4f}; rel {dn:.2e} <= {bn:.2e})")
        if dl > bl or dn > bn:
            bad.append((step, dl, bl, dn, bn))
    assert not bad, bad
    sd = net.state_dict()
    worst, worst_run, where, tight = 0.0, 0.0, "", 0.0
    env_w = dict(zip([str(k) for k in g["names"]], g["env_w"]))
    for k in g["names"]:
        k = str(k)
        a = sd[k].detach().cpu().contiguous().reshape(-1)
        got = (a if a.numel() <= 4096 else a[::max(1, a.numel() // 2048)]).numpy()
        ref = g["w__" + k]
        err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
        bound = C_ENV * float(env_w[k]) + (1e-4 if "running" in k else FLOOR_W)     # running statistics: the one-step fixtures' rtol
        tight = max(tight, err / bound)
        if "running" in k:
            worst_run = max(worst_run, err)
        elif err > worst:
            worst, where = err, k
        if err > bound:
            bad.append((k, err, bound))
    print(f"5-step trajectory: weights rel L-inf {worst:.2e} ({where}), running statistics {worst_run:.2e}; largest error / bound {tight:.2f}")
    assert not bad, bad
    assert int(sd["encoder.backbone.bn1.num_batches_tracked"]) == 5
