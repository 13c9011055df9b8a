"""Gradients of one training step with the discrete decisions FROZEN: a bound that can see a kernel bug.

The end-to-end comparison with the reference's gradients (test_train_gpu.py) has a noise floor of ~1e-3: the step switches
discretely -- ReLU signs, max-pool winners, the winning prototype -- and two float32 evaluations of the same step do not take
the same switches (one flipped arg-max of VGG-16's stride-1 pool moves every gradient below it by 1e-3).  Here the switches
are taken out of the comparison: the decisions the HIP forward pass actually took are read from its tape (ReLU: y > 0 of the
layer's output; max pool: the winners of the layer's float32 input, first maximum in scan order as the kernel and ATen both
choose; prototype maximum: the winners the cosine backward recorded in its workspace), and the oracle evaluates the SAME decision-frozen function in float64
under autograd (oracle/ref_cpu.py: Switches, frozen_gradients).  What is left between the two gradients is rounding only:

    every parameter tensor:  |hip - g64|_2 <= max(1e-5 * |g64|_2,  factor * |cpu32 - g64|_2)

where cpu32 is the same frozen function evaluated in float32 by the oracle on the CPU.

* The two Baselines stay under the absolute 1e-5 (measured: VGG-16 2e-6 .. 3.3e-6, ResNet-50 3e-6 .. 7.6e-6; their
  end-to-end bound is 3e-3).
* Stage 1's ENCODER (trunk, purifier, ASPPV2, DropBlock on and off) is taken alone under a linear probe -- the head
  replaced by sum(features * R) -- and held to factor 3 with a cap of 5e-5 (measured 1.2e-5 .. 2.5e-5, the oracle's float32
  1.2e-5 .. 2.6e-5: a white-noise R through 50 batch-statistics BatchNorm backward passes).  Stage 2's encoder (ResNetCM with
  its communication modules, ASPP, Dropout2d on and off) likewise: 8.7e-6 .. 9.1e-6 against 8.5e-6 .. 1.7e-5.
* Stage 1's FULL step is ill-conditioned in float32 whoever evaluates it: the meta-prototype head takes a softmax over
  -|x - c|^2 of 512-dimensional features with |x - c|^2 in the hundreds, so the 1e-5 relative error a float32 forward pass
  leaves in the features (and, in the reference's formulation, the 1e-4 absolute rounding of the distances themselves --
  the HIP head avoids that part, csrc/head_common.h) comes out as ~1e-4 in every gradient.  Measured over several runs: HIP
  0.9e-4 .. 4.4e-4, the oracle's float32 0.7e-4 .. 2.7e-4, in no fixed ratio (0.6 .. 3.8: the realised decisions differ from
  run to run with the autotuned tile choices).  Those two tests use factor 8: they pin the step's structure (every decision
  consumed, the DropBlock masks, the prototype routing) and a 1e-3-class error, not the last digit; the last digit is pinned
  by the probe test and by the per-kernel tests of the head (test_train_ops_gpu.py).

The loss is held to 2e-5 in the full-step cases: the bound the one-step fixture tests state for it (test_train_gpu.py; rounds 3-5
had 2e-6 here, 2.5 x the largest value seen until then -- with the kernel picks pinned the same step gives 2.9e-6: a float32
forward pass leaves ~1e-5 relative in the features, the cross-entropy averages that over 2 x 97 x 97 pixels)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
BOUND = 1e-5
FACTOR = 3.0
FULL_STEP_FACTOR = 8.0         # stage 1 through its float32-ill-conditioned head, see above


def _nchw(t, perm=None):
    t = t.detach().permute(0, 3, 1, 2).contiguous().cpu()
    return t if perm is None else t[perm]


def _perm(B, S, Q):
    """oracle image order (episode-major: s.., q.. of episode 0, then episode 1 ...) -> HIP image order ([all supports | all
    queries])."""
    idx = []
    for b in range(B):
        idx += [b * S + s for s in range(S)] + [B * S + b * Q + q for q in range(Q)]
    return torch.tensor(idx)


def _resnet_decisions(tape, perm, prefix="encoder.backbone"):
    d = {f"{prefix}.relu": _nchw(tape["stem"]["y"], perm) > 0}
    d[f"{prefix}.maxpool"] = F.max_pool2d(_nchw(tape["pool_in"], perm), 3, 2, 1, ceil_mode=True, return_indices=True)[1]
    names = [(L, i) for L, n in (("layer1", 3), ("layer2", 4), ("layer3", 6)) for i in range(n)]
    assert len(names) == len(tape["blocks"])
    for (L, i), rec in zip(names, tape["blocks"]):
        for r in (1, 2, 3):
            d[f"{prefix}.{L}.{i}.relu{r}"] = _nchw(rec[f"r{r}"]["y"], perm) > 0
    return d


def _stage1_tail_decisions(tape, perm, midc, prefix="encoder.purifier"):
    d = {f"{prefix}.0.relu": _nchw(tape["ya"], perm) > 0, f"{prefix}.3.relu": _nchw(tape["yb"], perm) > 0,
         f"{prefix}.6.aspp_0.relu": _nchw(tape["g0"], perm) > 0}
    for i in range(1, 5):
        d[f"{prefix}.6.aspp_{i}.relu"] = _nchw(tape["cat"][..., (i - 1) * midc:i * midc], perm) > 0
    return d


def _vgg_decisions(tape, perm, prefix="encoder.backbone"):
    from pemp_amd.networks.backbones import VGG_LAYOUT
    d, npool, prev = {}, 0, None
    convs = iter([it for it in VGG_LAYOUT if isinstance(it, tuple)])
    for kind, obj, relu, x, y in tape["vgg"]:
        if kind == "conv":
            idx = next(convs)[0]
            if relu:
                d[f"{prefix}.features.{idx}.relu"] = _nchw(y, perm) > 0
            prev = y
        else:
            d[f"{prefix}.pool{npool}"] = F.max_pool2d(_nchw(prev, perm), 3, obj, 1, return_indices=True)[1]
            npool += 1
            prev = None       # the pooled tensor is the next conv's input; only conv outputs are needed here
    return d


def _stage2_decisions(tape, perm, midc):
    """ResNetCM trunk (+ the arg-max pixel of each communication module's masked maximum, per image and channel) and the
    stage-2 purifier / ASPP (conv -> ReLU -> Dropout2d: the tape holds the ReLU outputs)."""
    d = _resnet_decisions(tape, perm)
    for i, cm in enumerate(tape["cm"]):
        d[f"encoder.backbone.linear{i + 1}.max"] = cm["arg"].detach().cpu().long()[perm]
    p = "encoder.purifier"
    d.update({f"{p}.0.relu": _nchw(tape["ya"], perm) > 0, f"{p}.3.relu": _nchw(tape["yb"], perm) > 0,
              f"{p}.6.aspp_0.relu": _nchw(tape["g0"], perm) > 0})
    for i in range(1, 5):
        d[f"{p}.6.aspp_{i}.relu"] = _nchw(tape["cat"][..., (i - 1) * midc:i * midc], perm) > 0
    return d


def _run(tr, net, batch, model, backbone, tail, probe=False, extra=()):
    """-> (hip loss, {name: hip gradient}, decisions of the HIP forward pass[, the probe R in the oracle's layout]).
    ``probe``: the head is replaced by loss = sum(features * R) with a random R, so d loss / d features = R exactly."""
    from oracle import ref_cpu
    sup, msk, qry, gt = batch
    B, S = sup.shape[:2]
    Q = qry.shape[1]
    perm = _perm(B, S, Q)
    grabbed = {}
    eng = tr.eng
    orig_backward, orig_head = eng.backward, tr._head_hip

    def spy_backward(dfeat):
        grabbed["tape"] = dict(eng.tape)
        return orig_backward(dfeat)

    def spy_head(feat, *a):
        grabbed["feat"] = feat.detach().clone()
        if not probe:
            return orig_head(feat, *a)
        gen = torch.Generator().manual_seed(5)
        R = (torch.randn(feat.shape, generator=gen) / feat.shape[0] ** 0.5).to(feat.device)     # NHWC, the engine's image order
        grabbed["R"] = R
        loss = (feat.double() * R.double()).sum()
        eng.backward(R.clone())
        return loss, None

    eng.backward, tr._head_hip = spy_backward, spy_head
    try:
        loss, _ = tr.forward_backward(sup, msk, qry, gt, *extra)
    finally:
        eng.backward, tr._head_hip = orig_backward, orig_head
    torch.cuda.synchronize()
    tape = grabbed["tape"]
    if model == "stage2":
        dec = _stage2_decisions(tape, perm, eng.midc)
    else:
        dec = _vgg_decisions(tape, perm) if backbone == "vgg16" else _resnet_decisions(tape, perm)
    if tail and model != "stage2":
        dec.update(_stage1_tail_decisions(tape, perm, eng.midc))
    if model == "stage1" and not probe:
        # winning prototypes: what the cosine backward itself routed every (query pixel, group) gradient to -- it leaves them
        # in the tail of its workspace (pemp_head_bwd_workspace_bytes).  Coinciding meta-prototypes give exact ties over whole
        # regions, so a recomputation in other arithmetic would not reproduce these choices.
        from pemp_amd import _lib
        feat = grabbed["feat"]
        c, h, w = feat.shape[3], feat.shape[1], feat.shape[2]
        p = net.ctr.shape[1] // 2
        n = h * w
        nbytes = _lib.load().pemp_head_bwd_workspace_bytes(B, S, n, c, p)
        ws = eng.ws[("head_bwd", B, S, h, w, c, p)]
        win = ws[nbytes - B * 2 * n * 4:nbytes].view(torch.int32).view(B, 2, h, w).cpu().long()
        assert int(win[:, 0].min()) >= 0 and int(win[:, 0].max()) < p and int(win[:, 1].min()) >= p and int(win[:, 1].max()) < 2 * p
        # oracle layout (compute_similarity): channel 0 = background, 1 = foreground, index inside the group
        dec["head.proto_max"] = torch.stack((win[:, 1] - p, win[:, 0]), dim=1)
    grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.requires_grad and p.grad is not None}
    if probe:
        grads.pop("ctr", None)                                         # the head's parameter: no gradient from the probe
        return float(loss.item()), grads, dec, _nchw(grabbed["R"], perm)
    return float(loss.item()), grads, dec


def _compare(what, hip_loss, hip, sd, batch, dec, model, backbone, dropblock=None, probe=None, factor=FACTOR, loss_rtol=2e-5,
             **more):
    from oracle import ref_cpu
    sup, msk, qry, gt = (t.cpu() for t in batch)
    kw = dict(model=model, backbone=backbone, dropblock=dropblock, probe=probe, **more)
    loss64, g64, used = ref_cpu.frozen_gradients(sd, sup, msk, qry, gt, dec, **kw)
    assert used == set(dec), (sorted(set(dec) - used), sorted(used - set(dec)))      # every decision of the pass was frozen
    # the same frozen function in float32 on the CPU: what float32 arithmetic itself leaves of the float64 gradient
    _, g32, _ = ref_cpu.frozen_gradients(sd, sup, msk, qry, gt, dec, dtype=torch.float32, **kw)
    assert loss_rtol is None or abs(hip_loss - loss64) <= loss_rtol * max(1.0, abs(loss64)), (hip_loss, loss64)
    assert set(hip) == set(g64), sorted(set(hip) ^ set(g64))[:10]
    rows = []
    top = max(g.norm().item() for g in g64.values())
    for name, g in g64.items():
        n2 = g.norm().item()
        if n2 <= 1e-10 * top:          # an exact zero (stage 2: the bias of a communication module's Linear shifts a channel that is
            assert hip[name].double().norm().item() <= 1e-6 * top, name      # constant over the batch; the next BatchNorm removes it)
            continue
        rows.append(((hip[name].double() - g).norm().item() / n2, (g32[name].double() - g).norm().item() / n2, name))
    rows.sort(reverse=True)
    med = rows[len(rows) // 2]
    print(f"{what}: {len(rows)} tensors, relative L2 error vs float64: hip max {rows[0][0]:.2e} ({rows[0][2]}), median {med[0]:.2e}; "
          f"cpu float32 max {max(r[1] for r in rows):.2e}, median {sorted(r[1] for r in rows)[len(rows) // 2]:.2e}")
    for e_hip, e_32, name in rows[:5]:
        print(f"   {name:50s} hip {e_hip:.2e}   cpu float32 {e_32:.2e}")
    bad = [(n, eh, e32) for eh, e32, n in rows if eh > max(BOUND, factor * e32)]
    assert not bad, (what, bad[:10])
    return rows


def _batch(dev, seeds=(31, 32), H=97):
    from pemp_amd import synth
    b = synth.make_batch(list(seeds), shot=1, height=H, width=H, out_hw=(H, H))
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])


@pytest.mark.parametrize("rate", [0.0, 0.5])
def test_stage2_encoder_gradients_under_a_linear_probe(hip_lib, dev, rate):
    """The stage-2 ENCODER (4-channel stem, ResNetCM with its three communication modules -- masked mean / max statistics,
    episode mean, Linear(2C -> 2), the two constant channels as a per-image conv bias -- trainable block BNs, purifier and
    ASPP with Dropout2d off / on with given draws) under the linear probe: every ReLU, the max pool, and the arg-max pixel of
    each communication module's maximum are replayed by the oracle in float64.  Bound: factor 3 on the oracle's float32
    error, capped at 5e-5.  (The end-to-end stage-2 comparisons sit at the 3e-3 switching floor.)"""
    from oracle import ref_cpu
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage2 as m
    from pemp_amd.train_stage2 import Stage2Trainer
    from tests.golden.cases import stage2_train_prior
    sd = util.wgen_state_dict("stage2_rn50cm", seed=4321)
    net = m.ModelClass(1, 1, None)
    net.load_state_dict(sd)
    tr = Stage2Trainer(None, net, device=dev, drop_rate2=rate)
    batch = _batch(dev)
    b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"]))
    do = None
    if rate > 0:
        perm = _perm(2, 1, 1)
        layers = ["encoder.purifier.2", "encoder.purifier.5"] + [f"encoder.purifier.6.aspp_{i}.2" for i in range(5)]
        gen = torch.Generator().manual_seed(79)
        draws = {k: torch.rand((4, 256), generator=gen) for k in layers}
        tr.eng.draws = {k: v.to(dev) for k, v in draws.items()}
        do = ref_cpu.Dropout2d(rate, {k: v[perm] for k, v in draws.items()})
    hip_loss, hip, dec, R = _run(tr, net, batch, "stage2", "resnet50", tail=True, probe=True, extra=(prior.to(dev),))
    rows = _compare(f"stage2_rn50cm encoder, probe, drop_rate2 {rate}", hip_loss, hip, sd, batch, dec, "stage2", "resnet50",
                    probe=R, loss_rtol=None, qry_prior=prior, dropout2d=do)
    assert rows[0][0] <= 5e-5
    assert do is None or len(do.used) == 7


def test_stage1_rn50_gradients_with_frozen_decisions(hip_lib, dev):
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    tr = Stage1Trainer(net, device=dev, drop_rate=0.0)
    batch = _batch(dev)
    hip_loss, hip, dec = _run(tr, net, batch, "stage1", "resnet50", tail=True)
    _compare("stage1_rn50", hip_loss, hip, sd, batch, dec, "stage1", "resnet50", factor=FULL_STEP_FACTOR)


def _dropblock_case(dev, rate, bs, B=2, S=1, Q=1, H=97):
    """-> ({layer: draws on the device, engine image order}, oracle DropBlock with the same draws in its image order)."""
    from oracle import ref_cpu
    nimg, perm = B * (S + Q), _perm(B, S, Q)
    h = w = (H - 1) // 8 + 1
    layers = {"encoder.purifier.2": (h, w), "encoder.purifier.5": (h, w), "encoder.purifier.6.aspp_0.1": (1, 1)}
    layers.update({f"encoder.purifier.6.aspp_{i}.1": (h, w) for i in range(1, 5)})
    gen = torch.Generator().manual_seed(77)
    draws = {k: torch.rand((nimg,) + hw, generator=gen) for k, hw in layers.items()}
    # the 1 x 1 global branch drops a whole image with probability rate / block^2: make sure one such drop is in the case
    draws["encoder.purifier.6.aspp_0.1"][1] = 0.0
    assert all(bool((v < rate / bs ** 2).any()) for v in draws.values())         # every layer drew at least one seed
    return {k: v.to(dev) for k, v in draws.items()}, ref_cpu.DropBlock(rate, bs, {k: v[perm] for k, v in draws.items()})


@pytest.mark.parametrize("rate", [0.0, 0.1])
def test_stage1_encoder_gradients_under_a_linear_probe(hip_lib, dev, rate):
    """The stage-1 ENCODER alone (ResNet-50 trunk, purifier, ASPPV2 with its five batch-statistics BNs and -- rate 0.1 -- the
    seven DropBlock layers): the prototype head, whose float32 conditioning sets the ~1e-4 floor of the full-step tests
    below, is replaced on both sides by the linear functional sum(features * R), so the gradient entering the encoder's
    backward pass is R exactly.  Bound: factor 3 on the oracle's float32 error, capped at 5e-5."""
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    tr = Stage1Trainer(net, device=dev, drop_rate=rate, block_size=4)
    db = None
    if rate > 0:
        tr.eng.draws, db = _dropblock_case(dev, rate, 4)
    batch = _batch(dev)
    hip_loss, hip, dec, R = _run(tr, net, batch, "stage1", "resnet50", tail=True, probe=True)
    rows = _compare(f"stage1_rn50 encoder, probe, drop_rate {rate}", hip_loss, hip, sd, batch, dec, "stage1", "resnet50",
                    dropblock=db, probe=R, loss_rtol=None)      # the probe's value is a sum of 3.5e5 signed terms: no test
    assert rows[0][0] <= 5e-5
    assert db is None or len(db.used) == 7


def test_stage1_rn50_gradients_with_dropblock_active(hip_lib, dev):
    """The full step with the regulariser ON (the shipped configuration: drop_rate 0.1, block_size 4): the seven DropBlock2D
    layers of the purifier / ASPPV2 get the SAME uniform draws on both sides -- the engine through ``eng.draws``, the oracle
    through its restatement of dropblock==0.3.0 (oracle/ref_cpu.py: DropBlock) -- so the block masks, their
    numel / sum normalisation over all images of the step, and their place in the backward pass are all inside the comparison
    (image order differs between the two sides, the draws are permuted with it)."""
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    rate, bs = 0.1, 4
    tr = Stage1Trainer(net, device=dev, drop_rate=rate, block_size=bs)
    tr.eng.draws, db = _dropblock_case(dev, rate, bs)
    batch = _batch(dev)
    hip_loss, hip, dec = _run(tr, net, batch, "stage1", "resnet50", tail=True)
    _compare("stage1_rn50 + DropBlock", hip_loss, hip, sd, batch, dec, "stage1", "resnet50", dropblock=db, factor=FULL_STEP_FACTOR)
    assert len(db.used) == 7
    # and the draws mattered: the loss differs from the regulariser-free step of the same batch
    tr0 = Stage1Trainer(net, device=dev, drop_rate=0.0)
    loss0, _ = tr0.forward_backward(*batch)
    assert abs(float(loss0.item()) - hip_loss) > 1e-4


@pytest.mark.parametrize("backbone,tag", [("vgg16", "baseline_vgg16"), ("resnet50", "baseline_rn50")])
def test_baseline_gradients_with_frozen_decisions(hip_lib, dev, backbone, tag):
    from pemp_amd.networks import baseline as m
    from pemp_amd.train_baseline import BaselineTrainer
    sd = util.wgen_state_dict(tag)
    net = m.Baseline(None, backbone=backbone)
    net.load_state_dict(sd)
    tr = BaselineTrainer(net, device=dev)
    batch = _batch(dev)
    hip_loss, hip, dec = _run(tr, net, batch, "baseline", backbone, tail=False)
    _compare(tag, hip_loss, hip, sd, batch, dec, "baseline", backbone)


def test_stage1_encoder_gradients_under_a_linear_probe_at_the_training_shape(hip_lib, dev):
    """The same frozen-decision probe at BASELINE.json configs[2]'s per-rank shape -- FOUR episodes of 401 x 401 (8 images,
    20 808 feature rows: the 128 x 128 / split-K conv variants, the split-M weight gradients with their second pass, the
    statistics epilogues over 163 row tiles) -- instead of 2 x 97 x 97: every ReLU sign and the max-pool winners of the HIP
    pass are replayed by the oracle in float64 and in float32 on the host (about a minute of CPU work), the same
    max(1e-5, 3 x cpu32) rule per parameter tensor, cap 5e-5."""
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    tr = Stage1Trainer(net, device=dev, drop_rate=0.0)
    batch = _batch(dev, seeds=(1234, 1235, 1236, 1237), H=401)
    hip_loss, hip, dec, R = _run(tr, net, batch, "stage1", "resnet50", tail=True, probe=True)
    del tr
    torch.cuda.empty_cache()
    rows = _compare("stage1_rn50 encoder, probe, 4 x 401 x 401", hip_loss, hip, sd, batch, dec, "stage1", "resnet50",
                    probe=R, loss_rtol=None)
    assert rows[0][0] <= 5e-5
