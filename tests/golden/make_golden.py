#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never has it).  The
reference's hot-path modules (networks/backbones.py, pemp_stage1.py, pemp_stage2.py,
baseline.py, panet.py) are imported unmodified from /root/reference and executed on CPU.  Two third-party
packages they import are not installed in the image (no network): ``sacred`` (config injection)
and ``dropblock`` (train-only regulariser).  This script puts two minimal in-process stand-ins
for those packages on sys.path -- an ``Ingredient`` whose ``capture`` fills missing arguments
by name from a dict, and a ``DropBlock2D`` that is the identity (its eval-mode behaviour) --
exactly as SURVEY.md §8(c) recorded.  No reference source is copied or altered.

Inputs come from pemp_amd.synth (bit-exact on any host), so each fixture stores only the seeds
and the expected OUTPUTS (+ a few sampled intermediates).

usage:  python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import json
import logging
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from pemp_amd import synth  # noqa: E402
from tests.golden.cases import cedt_cases, metric_cases, metric_cases_coco, stage2_train_prior  # noqa: E402


def _install_standins():
    import inspect
    import functools

    sacred = types.ModuleType("sacred")

    class Ingredient:
        def __init__(self, name, **kw):
            self.name = name
            self.cfg = {}

        def config(self, fn):
            return fn

        def config_hook(self, fn):
            return fn

        def capture(self, fn):
            sig = inspect.signature(fn)
            ing = self

            @functools.wraps(fn)
            def wrapper(*a, **k):
                bound = sig.bind_partial(*a, **k)
                for p in sig.parameters:
                    if p not in bound.arguments and p in ing.cfg:
                        k[p] = ing.cfg[p]
                return fn(*a, **k)
            return wrapper

    sacred.Ingredient = Ingredient
    sys.modules["sacred"] = sacred

    dropblock = types.ModuleType("dropblock")

    class DropBlock2D(torch.nn.Module):
        def __init__(self, drop_prob=0.1, block_size=4):
            super().__init__()

        def forward(self, x):
            return x        # eval behaviour; in train mode this equals drop_prob = 0

    dropblock.DropBlock2D = DropBlock2D
    sys.modules["dropblock"] = dropblock


def _build(module, cls_name, cfg, ctor_args, tmp):
    """Instantiate a reference model with random torchvision-layout 'pretrained' file, eval mode."""
    from networks import backbones
    module.net_ingredient.cfg = dict(cfg)
    bb = cfg.get("backbone2") if cls_name == "PEMPStage2" else cfg["backbone"]
    if bb == "vgg16":
        pre = backbones.VGG16(3, None).state_dict()
    else:
        pre = backbones.ResNet(3, backbones.BottleNeck, [3, 4, 6, 3], pretrained=None).state_dict()
    f = Path(tmp) / f"pre_{bb}.pth"
    torch.save(pre, f)
    module.pretrained_weights[bb] = f
    logger = logging.getLogger("golden")
    model = getattr(module, cls_name)(*ctor_args, logger)
    return model.eval()


def _load_wgen(model, seed=1234):
    sd = synth.gen_state_dict({k: v for k, v in model.state_dict().items()}, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def _keys_fixture(model, name):
    spec = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    (OUT / f"state_keys_{name}.json").write_text(json.dumps(spec))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _metric_counts(pred, ref):
    out = []
    for j in (0, 1):
        v = ref != 255
        out.append([int(((pred == j) & (ref == j) & v).sum()),
                    int(((pred == j) & (ref != j) & v).sum()),
                    int(((pred != j) & (ref == j) & v).sum())])
    return np.array(out, np.int64)


def _capture_lowres(model, mod):
    """Hook compute_similarity/mpm to grab features-level intermediates without touching the reference."""
    grabbed = {}
    enc = model.encoder

    def hook(_m, _i, o):
        grabbed["features"] = o.detach()
    h = enc.register_forward_hook(hook)
    return grabbed, h


def _episode_case(model, kind, seeds, shot, H, out_hws, extra=None):
    """Run reference on episodes; return dict of arrays."""
    res = {"seeds": np.array(seeds), "shot": np.array(shot), "H": np.array(H)}
    grabbed, h = _capture_lowres(model, None)
    for n, seed in enumerate(seeds):
        ep = synth.make_episode(seed, shot=shot, height=H, width=H, out_hw=out_hws[n])
        sup, msk, qry = _t(ep["sup_img"])[None], _t(ep["sup_mask"])[None], _t(ep["qry_img"])[None]
        gt = _t(ep["qry_mask"])
        out_shape = tuple(gt.shape[-2:])
        with torch.no_grad():
            if kind == "stage1":
                logits, resp = model(sup, msk, qry, out_shape, ret_ind=True)
            elif kind == "baseline":
                logits, resp = model(sup, msk, qry, out_shape), None
            elif kind == "panet":
                (logits, aux), resp = model(sup, msk, qry, out_shape), None
                res[f"e{n}_align_loss"] = np.array(float(aux), np.float64)
            elif kind == "stage2":
                prior = extra["prior"][n]
                logits, resp = model(sup, msk, qry, prior, out_shape, ret_ind=True)
            loss = float(torch.nn.functional.cross_entropy(logits, gt, ignore_index=255))
        feats = grabbed["features"]                       # [S+Q, c, h, w]
        pred = logits.argmax(1).numpy().astype(np.uint8)
        res[f"e{n}_out_hw"] = np.array(out_shape)
        res[f"e{n}_loss"] = np.array(loss, np.float64)
        res[f"e{n}_argmax_bits"] = np.packbits(pred.reshape(-1))
        res[f"e{n}_counts"] = _metric_counts(pred[0], ep["qry_mask"][0])
        res[f"e{n}_logits_s7"] = logits[0, :, ::7, ::7].numpy()
        res[f"e{n}_feat_c8"] = feats[:, ::8].numpy() if feats.shape[-1] <= 13 else feats[:, ::8, ::5, ::5].numpy()
        if resp is not None:
            res[f"e{n}_resp_s7"] = resp[0, ::7, ::7].numpy().astype(np.uint8)
        if kind == "stage2":
            res[f"e{n}_adaptive_p"] = model.adaptive_p.numpy()
        if H <= 97:
            res[f"e{n}_logits"] = logits[0].numpy()
    h.remove()
    return res


def gen_stage1(tmp, backbone, tag, cases):
    from networks import pemp_stage1 as m
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone=backbone, protos=3,
               drop_rate=0.1, block_size=4)
    model = _build(m, "PEMPStage1", cfg, (), tmp)
    _load_wgen(model)
    _keys_fixture(model, tag)
    for cname, (seeds, shot, H, hws) in cases.items():
        np.savez_compressed(OUT / f"{tag}_{cname}.npz", **_episode_case(model, "stage1", seeds, shot, H, hws))
        print("wrote", tag, cname)
    return model


def gen_stage1_map(tmp):
    """protos=0 branch: plain masked average pooling (pemp_stage1.py:223-228)."""
    from networks import pemp_stage1 as m
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=0,
               drop_rate=0.1, block_size=4)
    model = _build(m, "PEMPStage1", cfg, (), tmp)
    _load_wgen(model)
    grabbed, h = _capture_lowres(model, None)
    res = {}
    ep = synth.make_episode(11, shot=2, height=97, width=97, out_hw=(80, 120))
    with torch.no_grad():
        logits = model(_t(ep["sup_img"])[None], _t(ep["sup_mask"])[None], _t(ep["qry_img"])[None], (80, 120))
    res["seeds"] = np.array([11]); res["shot"] = np.array(2); res["H"] = np.array(97)
    res["e0_out_hw"] = np.array((80, 120)); res["e0_logits"] = logits[0].numpy()
    np.savez_compressed(OUT / "stage1_rn50_map_small.npz", **res)
    h.remove()
    print("wrote stage1 map")


def gen_baseline(tmp, backbone, tag, cases):
    from networks import baseline as m
    cfg = dict(dist_scalar=20, init_channels=3, backbone=backbone, out_channels=512)
    model = _build(m, "Baseline", cfg, (), tmp)
    _load_wgen(model)
    _keys_fixture(model, tag)
    for cname, (seeds, shot, H, hws) in cases.items():
        np.savez_compressed(OUT / f"{tag}_{cname}.npz", **_episode_case(model, "baseline", seeds, shot, H, hws))
        print("wrote", tag, cname)


def gen_stage2(tmp, stage1_model, cases):
    from networks import pemp_stage2 as m
    for cname, (seeds, shot, H, hws) in cases.items():
        cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3,
                   drop_rate=0.1, block_size=4, backbone2="resnet50", protos2=3, drop_rate2=0.5, cm=True)
        model = _build(m, "PEMPStage2", cfg, (shot, 1), tmp)
        _load_wgen(model, seed=4321)
        _keys_fixture(model, "stage2_rn50cm")
        priors = []
        for n, seed in enumerate(seeds):                    # stage-1 prior, entry/pemp_stage2.py:58-60
            ep = synth.make_episode(seed, shot=shot, height=H, width=H, out_hw=hws[n])
            with torch.no_grad():
                p = stage1_model(_t(ep["sup_img"])[None], _t(ep["sup_mask"])[None], _t(ep["qry_img"])[None])
            priors.append(p.argmax(dim=1, keepdim=True))
        res = _episode_case(model, "stage2", seeds, shot, H, hws, extra={"prior": priors})
        for n, p in enumerate(priors):
            res[f"e{n}_prior_bits"] = np.packbits(p.numpy().astype(np.uint8).reshape(-1))
        np.savez_compressed(OUT / f"stage2_rn50cm_{cname}.npz", **res)
        print("wrote stage2", cname)


TRAIN_FULL = dict(seeds=(1234, 1235, 1236, 1237), H=401, out="stage1_rn50_trainstep_full")     # BASELINE.json configs[2]


def gen_stage2_vgg(tmp, stage1_model, cases):
    """G23: stage 2 on VGG16CM (networks/backbones.py:424-533).  As shipped the class cannot be built through
    PEMPStage2 with a pretrained file (init_weights reads an attribute that is never set, :518); with
    ``pretrained_weights["vgg16"] = None`` the unmodified constructor runs (no import step), which is what this does."""
    from networks import pemp_stage2 as m
    for cname, (seeds, shot, H, hws) in cases.items():
        cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3,
                   drop_rate=0.1, block_size=4, backbone2="vgg16", protos2=3, drop_rate2=0.5, cm=True)
        m.net_ingredient.cfg = dict(cfg)
        m.pretrained_weights["vgg16"] = None
        model = m.PEMPStage2(shot, 1, logging.getLogger("golden")).eval()
        _load_wgen(model, seed=4321)
        _keys_fixture(model, "stage2_vgg16cm")
        priors = []
        for n, seed in enumerate(seeds):
            ep = synth.make_episode(seed, shot=shot, height=H, width=H, out_hw=hws[n])
            with torch.no_grad():
                p = stage1_model(_t(ep["sup_img"])[None], _t(ep["sup_mask"])[None], _t(ep["qry_img"])[None])
            priors.append(p.argmax(dim=1, keepdim=True))
        res = _episode_case(model, "stage2", seeds, shot, H, hws, extra={"prior": priors})
        for n, p in enumerate(priors):
            res[f"e{n}_prior_bits"] = np.packbits(p.numpy().astype(np.uint8).reshape(-1))
        np.savez_compressed(OUT / f"stage2_vgg16cm_{cname}.npz", **res)
        print("wrote stage2 vgg16cm", cname)


def gen_train_step(tmp, seeds=(31, 32), H=97, out="stage1_rn50_trainstep"):
    """G9: one training step's loss and gradients of the reference in train() mode (batch-stat BN;
    DropBlock = identity, i.e. drop_rate 0), B=2 episodes, 97x97, CE loss (entry/pemp_stage1.py:57-65).
    G20 (``TRAIN_FULL``): the same at the shape BASELINE.json configs[2] trains at -- data.bs = 4 episodes, 401x401."""
    from networks import pemp_stage1 as m
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3,
               drop_rate=0.0, block_size=4)
    model = _build(m, "PEMPStage1", cfg, (), tmp)
    _load_wgen(model)
    model.train()
    b = synth.make_batch(list(seeds), shot=1, height=H, width=H, out_hw=(H, H))
    sup, msk, qry = _t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"])
    gt = _t(b["qry_mask"][:, 0])
    logits = model(sup, msk, qry, (H, H))
    loss = torch.nn.functional.cross_entropy(logits, gt, ignore_index=255)
    loss.backward()
    res = {"loss": np.array(float(loss), np.float64), "logits_s7": logits.detach()[:, :, ::7, ::7].numpy(),
           "seeds": np.array(seeds), "H": np.array(H)}
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms, np.float64)
    for k in ("ctr", "encoder.backbone.conv1.weight", "encoder.purifier.6.layer6.bias",
              "encoder.backbone.layer3.5.bn3.weight", "encoder.backbone.layer1.0.downsample.0.weight",
              "encoder.purifier.6.aspp_3.2.weight", "encoder.backbone.layer2.0.conv1.weight"):
        g = dict(model.named_parameters())[k].grad
        res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
    sd = model.state_dict()
    for k in ("encoder.backbone.bn1.running_mean", "encoder.backbone.layer3.5.bn3.running_var",
              "encoder.purifier.6.aspp_0.0.running_var", "encoder.backbone.bn1.num_batches_tracked"):
        res["buf__" + k] = sd[k].numpy()
    np.savez_compressed(OUT / f"{out}.npz", **res)
    print("wrote train step", out, "; loss", float(loss))


TRAJ_STEPS = 5


def traj_seeds(step):
    return (31 + 2 * step, 32 + 2 * step)


TRAJ_REPLICAS = 8          # perturbed replicas of the trajectory (the fixture's own sensitivity to one float32 ulp)
TRAJ_REL_EPS = 1e-7


def _sampled(v):
    a = v.detach().reshape(-1)
    return (a if a.numel() <= 4096 else a[::max(1, a.numel() // 2048)]).numpy()


def _run_trajectory(m, tmp, perturb_seed=None):
    """The reference's five steps from Wgen(1234); ``perturb_seed``: every parameter multiplied by 1 + 1e-7 N(0,1) first."""
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3, drop_rate=0.0, block_size=4)
    model = _build(m, "PEMPStage1", cfg, (), tmp)
    _load_wgen(model)
    if perturb_seed is not None:
        from tests import util
        util.perturb_parameters(model.named_parameters(), perturb_seed, TRAJ_REL_EPS)
    model.train()
    opt = torch.optim.SGD(model.parameters(), 1e-3, momentum=0.9, weight_decay=5e-4, nesterov=False)
    losses, norms = [], []
    for step in range(TRAJ_STEPS):
        b = synth.make_batch(list(traj_seeds(step)), shot=1, height=97, width=97, out_hw=(97, 97))
        opt.zero_grad()
        logits = model(_t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), (97, 97))
        loss = torch.nn.functional.cross_entropy(logits, _t(b["qry_mask"][:, 0]), ignore_index=255)
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.1)))
        opt.step()
        losses.append(float(loss.detach()))
    return model, np.array(losses, np.float64), np.array(norms, np.float64)


def gen_train_trajectory(tmp):
    """G25: FIVE consecutive training steps of the reference (entry/pemp_stage1.py:57-65 repeated by core/base_trainer.py:194-200):
    the imported PEMPStage1 in train() mode (DropBlock off), torch.optim.SGD as core/solver.py:87-91 builds it (lr 1e-3,
    momentum 0.9, weight decay 5e-4), clip_grad_norm_(1.1), a different batch of two 97 x 97 episodes per step.  Stored: the
    loss of every step, the gradient norm before clipping, and the weights / BatchNorm buffers after the last step.

    ENVELOPE (``env_*``): the same five steps by the same reference code from TRAJ_REPLICAS copies of the initial weights, every
    parameter multiplied by 1 + 1e-7 N(0,1) (about one float32 ulp; seeds 9000 + r).  Per step the largest |loss_r - loss| and
    |norm_r - norm| / norm over the replicas, per tensor the largest relative L-inf distance of the sampled final weights:
    how far ANY float32 evaluation of this trajectory may sit from the stored one.  The trajectory amplifies rounding from
    step to step (the gradient norm jumps 15 -> 37.5 at step 4), so a bound on a later step has to come from here."""
    from networks import pemp_stage1 as m
    model, losses, norms = _run_trajectory(m, tmp)
    res = {"losses": losses, "grad_norms": norms, "steps": np.array(TRAJ_STEPS)}
    sd = model.state_dict()
    names = []
    for k, v in sd.items():
        if not v.is_floating_point():
            res["buf__" + k] = v.numpy()
            continue
        names.append(k)
        res["w__" + k] = _sampled(v)
        res["norm__" + k] = np.array(float(v.detach().double().norm()), np.float64)
    res["names"] = np.array(names)
    env_l, env_n = np.zeros(TRAJ_STEPS), np.zeros(TRAJ_STEPS)
    env_w = {k: 0.0 for k in names}
    rep_l, rep_n = [], []
    for r in range(TRAJ_REPLICAS):
        mr, lr_, nr = _run_trajectory(m, tmp, perturb_seed=9000 + r)
        rep_l.append(lr_)
        rep_n.append(nr)
        env_l = np.maximum(env_l, np.abs(lr_ - losses))
        env_n = np.maximum(env_n, np.abs(nr - norms) / norms)
        sdr = mr.state_dict()
        for k in names:
            ref = res["w__" + k]
            env_w[k] = max(env_w[k], float(np.abs(_sampled(sdr[k]) - ref).max() / (np.abs(ref).max() + 1e-12)))
        print(f"  replica {r}: d loss {np.abs(lr_ - losses)}, rel d norm {np.abs(nr - norms) / norms}")
    res["env_loss"], res["env_norm"] = env_l, env_n
    res["env_w"] = np.array([env_w[k] for k in names], np.float64)
    res["env_replica_losses"], res["env_replica_norms"] = np.array(rep_l), np.array(rep_n)
    res["env_rel_eps"], res["env_replicas"] = np.array(TRAJ_REL_EPS), np.array(TRAJ_REPLICAS)
    np.savez_compressed(OUT / "stage1_rn50_trajectory.npz", **res)
    print("wrote stage1_rn50_trajectory; losses", losses, "grad norms", norms)
    print("  envelope: loss", env_l, "norm", env_n, "weights max", max(env_w.values()))


def gen_train_step_5shot(tmp):
    """G16: 5-shot training steps (mean over shots in the MPM / the PANet alignment branch's expansion over S):
    stage-1 ResNet-50 and PANet VGG-16, B = 2 episodes x (5 + 1) images, 97x97."""
    from networks import pemp_stage1 as m1, panet as mp
    for tag in ("stage1_rn50", "panet_vgg16"):
        # the reference's PANet supports S > 1 only for one episode per batch (its compute_similarity views an expanded
        # tensor, panet.py:135-139; scripts/panet.sh trains with data.bs=1): B = 1 there, B = 2 for stage 1
        b = synth.make_batch([41, 42] if tag == "stage1_rn50" else [41], shot=5, height=97, width=97, out_hw=(97, 97))
        sup, msk, qry, gt = _t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), _t(b["qry_mask"][:, 0])
        if tag == "stage1_rn50":
            cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3, drop_rate=0.0, block_size=4)
            model = _build(m1, "PEMPStage1", cfg, (), tmp)
        else:
            cfg = dict(dist_scalar=20, init_channels=3, backbone="vgg16", out_channels=512)
            model = _build(mp, "PANet", cfg, (), tmp)
        _load_wgen(model)
        model.train()
        out = model(sup, msk, qry, (97, 97))
        logits, aux = out if isinstance(out, tuple) else (out, None)
        loss = torch.nn.functional.cross_entropy(logits, gt, ignore_index=255)
        (loss if aux is None else loss + aux).backward()
        res = {"loss": np.array(float(loss.detach()), np.float64)}
        if aux is not None:
            res["align_loss"] = np.array(float(aux.detach()), np.float64)
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        res["grad_names"], res["grad_norms"] = np.array(names), np.array(norms, np.float64)
        plist = dict(model.named_parameters())
        for k in [k for k in plist if plist[k].grad is not None][:2] + list(plist)[-2:]:
            g = plist[k].grad
            res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
        np.savez_compressed(OUT / f"{tag}_trainstep5.npz", **res)
        print("wrote", tag, "5-shot train step; loss", float(loss), "" if aux is None else float(aux))


def gen_panet(tmp, backbone, tag, cases):
    """G14: PANet (networks/panet.py:68-193) = the Baseline forward + the prototype-alignment loss."""
    from networks import panet as m
    cfg = dict(dist_scalar=20, init_channels=3, backbone=backbone, out_channels=512)
    model = _build(m, "PANet", cfg, (), tmp)
    _load_wgen(model)
    _keys_fixture(model, tag)
    for cname, (seeds, shot, H, hws) in cases.items():
        np.savez_compressed(OUT / f"{tag}_{cname}.npz", **_episode_case(model, "panet", seeds, shot, H, hws))
        print("wrote", tag, cname)


def gen_train_step_panet(tmp):
    """G15: losses and gradients of one PANet training step (entry/panet.py:103-110: loss + loss_coef * align_loss)."""
    from networks import panet as m
    for backbone, tag in (("vgg16", "panet_vgg16"), ("resnet50", "panet_rn50")):
        cfg = dict(dist_scalar=20, init_channels=3, backbone=backbone, out_channels=512)
        model = _build(m, "PANet", cfg, (), tmp)
        _load_wgen(model)
        model.train()
        b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
        logits, aux = model(_t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), (97, 97))
        loss = torch.nn.functional.cross_entropy(logits, _t(b["qry_mask"][:, 0]), ignore_index=255)
        (loss + aux * 1.0).backward()
        res = {"loss": np.array(float(loss.detach()), np.float64), "align_loss": np.array(float(aux.detach()), np.float64)}
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        res["grad_names"], res["grad_norms"] = np.array(names), np.array(norms, np.float64)
        plist = dict(model.named_parameters())
        for k in [k for k in plist if plist[k].grad is not None][:2] + list(plist)[-2:]:
            g = plist[k].grad
            res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
        np.savez_compressed(OUT / f"{tag}_trainstep.npz", **res)
        print("wrote", tag, "train step; loss", float(loss), "align", float(aux))


def gen_train_step_baseline(tmp):
    """G10: loss and gradients of one Baseline training step (entry/baseline.py:54-62), VGG16 and ResNet-50."""
    from networks import baseline as m
    for backbone, tag in (("vgg16", "baseline_vgg16"), ("resnet50", "baseline_rn50")):
        cfg = dict(dist_scalar=20, init_channels=3, backbone=backbone, out_channels=512)
        model = _build(m, "Baseline", cfg, (), tmp)
        _load_wgen(model)
        model.train()
        b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
        logits = model(_t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), (97, 97))
        loss = torch.nn.functional.cross_entropy(logits, _t(b["qry_mask"][:, 0]), ignore_index=255)
        loss.backward()
        res = {"loss": np.array(float(loss.detach()), np.float64)}
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
        res["grad_names"], res["grad_norms"] = np.array(names), np.array(norms, np.float64)
        plist = dict(model.named_parameters())
        for k in [k for k in plist if plist[k].grad is not None][:2] + list(plist)[-2:]:
            g = plist[k].grad
            res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
        np.savez_compressed(OUT / f"{tag}_trainstep.npz", **res)
        print("wrote", tag, "train step; loss", float(loss))


def gen_train_step_stage1_vgg(tmp):
    """G12: one training step of stage 1 on VGG-16 (no purifier, MPM head with learnable ctr), as G9."""
    from networks import pemp_stage1 as m
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="vgg16", protos=3, drop_rate=0.0, block_size=4)
    model = _build(m, "PEMPStage1", cfg, (), tmp)
    _load_wgen(model)
    model.train()
    b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
    logits = model(_t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), (97, 97))
    loss = torch.nn.functional.cross_entropy(logits, _t(b["qry_mask"][:, 0]), ignore_index=255)
    loss.backward()
    res = {"loss": np.array(float(loss.detach()), np.float64)}
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    res["grad_names"], res["grad_norms"] = np.array(names), np.array(norms, np.float64)
    plist = dict(model.named_parameters())
    for k in ("ctr", "encoder.backbone.features.0.weight", "encoder.backbone.features.28.bias",
              "encoder.backbone.features.14.weight"):
        g = plist[k].grad
        res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
    np.savez_compressed(OUT / "stage1_vgg16_trainstep.npz", **res)
    print("wrote stage-1 VGG16 train step; loss", float(loss.detach()))


def gen_train_step_stage2(tmp, shot=1, seeds=(31, 32), out="stage2_rn50cm_trainstep", H=97):
    """G11: loss and gradients of one stage-2 training step (entry/pemp_stage2.py:72-83): ResNet-50+CM in
    train() mode (batch-stat BN, Dropout2d off = drop_rate2 0), B=2 episodes, 97x97, CE loss.
    (shot=5: G17, the communication modules' episode means then run over S + Q = 6 images.  H=401: G22, one 5-shot
    episode at the full input size of BASELINE.json configs[3].)"""
    from networks import pemp_stage2 as m
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3,
               drop_rate=0.1, block_size=4, backbone2="resnet50", protos2=3, drop_rate2=0.0, cm=True)
    model = _build(m, "PEMPStage2", cfg, (shot, 1), tmp)
    _load_wgen(model, seed=4321)
    model.train()
    b = synth.make_batch(list(seeds), shot=shot, height=H, width=H, out_hw=(H, H))
    prior = torch.from_numpy(stage2_train_prior(b["qry_mask"]))
    logits = model(_t(b["sup_img"]), _t(b["sup_mask"]), _t(b["qry_img"]), prior, (H, H))
    loss = torch.nn.functional.cross_entropy(logits, _t(b["qry_mask"][:, 0]), ignore_index=255)
    loss.backward()
    res = {"loss": np.array(float(loss.detach()), np.float64), "logits_s7": logits.detach()[:, :, ::7, ::7].numpy(),
           "seeds": np.array(seeds), "H": np.array(H), "shot": np.array(shot)}
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    res["grad_names"], res["grad_norms"] = np.array(names), np.array(norms, np.float64)
    plist = dict(model.named_parameters())
    bb = "encoder.backbone."
    for k in ("ctr", bb + "conv1.weight", bb + "linear1.weight", bb + "linear2.bias", bb + "linear3.weight",
              bb + "layer1.0.conv1.weight", bb + "layer2.0.downsample.0.weight", bb + "layer3.0.conv1.weight",
              bb + "layer3.5.bn3.weight", bb + "layer1.1.bn2.bias", "encoder.purifier.6.aspp_3.0.weight",
              "encoder.purifier.6.aspp_0.0.bias", "encoder.purifier.6.layer6.bias", "encoder.purifier.0.weight"):
        g = plist[k].grad
        res["grad__" + k] = g.numpy() if g.numel() <= 40000 else g.reshape(-1)[::37].numpy()
    sd = model.state_dict()
    for k in (bb + "bn1.running_mean", bb + "layer3.5.bn3.running_var", bb + "layer2.0.downsample.1.running_mean",
              bb + "bn1.num_batches_tracked"):
        res["buf__" + k] = sd[k].numpy()
    np.savez_compressed(OUT / f"{out}.npz", **res)
    print("wrote stage-2 train step", out, "; loss", float(loss))


def gen_cedt():
    """G18: CELossDT of the reference itself (core/losses.py:17-43): weight maps and loss values.  The class is written
    for numpy < 1.24 and a CUDA box: ``np.bool`` is aliased to ``bool`` and ``Tensor.cuda`` is the identity while it
    runs here (environment shims in this process only; the reference file is imported unmodified)."""
    import core.losses as L
    had_bool = hasattr(np, "bool")
    if not had_bool:
        np.bool = bool
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        obj = L.CELossDT(5.0)
        res = {}
        seen = []
        inner = obj.boundary2weight
        obj.boundary2weight = lambda boundary: (seen.append(inner(boundary)), seen[-1])[1]   # record what __call__ computes
        for n, (tgt, logits) in enumerate(cedt_cases()):
            res[f"c{n}_loss"] = np.array(float(obj(logits, tgt)), np.float64)               # the reference's __call__
            res[f"c{n}_weight"] = seen[-1].numpy()                                           # ... and its weight map
    finally:
        torch.Tensor.cuda = orig_cuda
        if not had_bool:
            del np.bool
    np.savez_compressed(OUT / "cedt_reference.npz", **res)
    print("wrote cedt_reference", [float(res[f"c{n}_loss"]) for n in range(4)])


def gen_metric():
    """G19: FewShotMetric / Accumulator of the reference itself (core/metrics.py:4-66) on seeded predictions."""
    from core.metrics import Accumulator, FewShotMetric
    m = FewShotMetric(20)
    for pred, ref, cls in metric_cases():
        m.update(pred, ref, cls)
    labels = [1, 3, 5, 17]
    c, mean = m.mIoU(labels)
    cb, meanb = m.mIoU(labels, binary=True)
    acc = Accumulator(loss=[], miou=[], n=0.0)
    for i in range(3):
        acc.update(loss=0.5 + i, miou=c * (i + 1), n=2.0)
    np.savez_compressed(OUT / "metric_reference.npz", stat=m.stat, miou_c=c, miou=np.array(mean), biou_c=cb,
                        biou=np.array(meanb), acc_loss=np.array(acc.mean("loss")), acc_miou=acc.mean("miou", axis=0),
                        acc_n=np.array(acc.mean("n")))
    print("wrote metric_reference", float(mean), float(meanb))


def gen_metric_coco():
    """G21: the reference's FewShotMetric with 80 classes (entry/pemp_stage1.py:152) and COCO-20i validation labels
    (split 1: 21..40, the list data_kits/datasets.py:99-100 returns; that module needs torchvision / pycocotools and is
    not importable here, so the list is written out)."""
    from core.metrics import FewShotMetric
    m = FewShotMetric(80)
    for pred, ref, cls in metric_cases_coco():
        m.update(pred, ref, cls)
    labels = list(range(1 * 20 + 1, 1 * 20 + 21))
    c, mean = m.mIoU(labels)
    cb, meanb = m.mIoU(labels, binary=True)
    np.savez_compressed(OUT / "metric_reference_coco.npz", stat=m.stat, labels=np.array(labels), miou_c=c, miou=np.array(mean),
                        biou_c=cb, biou=np.array(meanb))
    print("wrote metric_reference_coco", m.stat.shape, float(mean), float(meanb))


REAL_DIR = REF / "http" / "static" / "1005_pascal_1shot_pemp_stage2_s0"


def gen_real_episodes(tmp, stage1_model):
    """G24: the two REAL PASCAL episodes the reference ships with its viewer (http/static/1005_pascal_1shot_pemp_stage2_s0/
    {000_01,001_03}: support / query JPEG, 0/255 PNG label images, data.json) through the reference's own pipeline:
    Pillow decode, the evaluation transform of data_kits/pascal_voc.py:200-229 (bilinear resize to 401 x 401, ToTensor,
    Normalize; support label nearest-resized, query label left at its size; that module needs torchvision, which is absent,
    so its four Pillow / tensor operations are written out here), then the imported PEMPStage1 -> argmax prior -> PEMPStage2
    with Wgen weights (entry/pemp_stage2.py:53-61).  Stored: the DECODED uint8 arrays (what Image.open yields; the GPU test
    feeds them to pemp_episode_preprocess), strided samples of the normalised tensors, and the outputs of both stages."""
    from PIL import Image
    from networks import pemp_stage2 as m2
    cfg = dict(dist_scalar=20, init_channels=3, out_channels=512, backbone="resnet50", protos=3,
               drop_rate=0.1, block_size=4, backbone2="resnet50", protos2=3, drop_rate2=0.5, cm=True)
    stage2 = _build(m2, "PEMPStage2", cfg, (1, 1), tmp)
    _load_wgen(stage2, seed=4321)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)            # data_kits/datasets.py:18-19
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    H = W = 401

    def image_tensor(pil):             # resize_image + ToTensor + Normalize (pascal_voc.py:141-144,201-202)
        a = np.asarray(pil.resize((W, H), Image.BILINEAR), np.uint8)
        t = torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().float().div(255)
        return (t - mean) / std

    res = {"dirs": np.array(sorted(d.name for d in REAL_DIR.iterdir() if d.is_dir()))}
    for n, name in enumerate(res["dirs"]):
        d = REAL_DIR / str(name)
        meta = json.loads((d / "data.json").read_text())
        cn = meta["cls_name"]
        sup_pil = Image.open(d / f"{cn}_sup_img_{meta['sup']}.jpg").convert("RGB")
        qry_pil = Image.open(d / f"{cn}_qry_img_{meta['qry']}.jpg").convert("RGB")
        sup_lab_pil = Image.open(d / f"{cn}_sup_msk_{meta['sup']}.png")
        qry_lab_pil = Image.open(d / f"{cn}_qry_msk_{meta['qry']}.png")
        sup_rgb, qry_rgb = image_tensor(sup_pil), image_tensor(qry_pil)
        sup_lab = np.asarray(sup_lab_pil.resize((W, H), Image.NEAREST), np.uint8)
        fg = torch.from_numpy((sup_lab // 255).astype(np.float32))
        sup_mask = torch.stack((fg, 1 - fg), dim=0)
        qry_lab = np.asarray(qry_lab_pil, np.uint8)
        gt = torch.from_numpy((qry_lab // 255).astype(np.int64))[None]
        sup, msk, qry = sup_rgb[None, None], sup_mask[None, None], qry_rgb[None, None]
        out_shape = tuple(gt.shape[-2:])
        with torch.no_grad():
            logits1, resp1 = stage1_model(sup, msk, qry, out_shape, ret_ind=True)
            prior = stage1_model(sup, msk, qry).argmax(dim=1, keepdim=True)           # entry/pemp_stage2.py:58-60
            logits2, resp2 = stage2(sup, msk, qry, prior, out_shape, ret_ind=True)
        e = f"e{n}_"
        res[e + "cls"] = np.array(int(meta["cls_id"]))
        res[e + "sup_img_u8"], res[e + "qry_img_u8"] = np.asarray(sup_pil, np.uint8), np.asarray(qry_pil, np.uint8)
        res[e + "sup_lab_u8"], res[e + "qry_lab_u8"] = np.asarray(sup_lab_pil, np.uint8), qry_lab
        res[e + "sup_rgb_s5"], res[e + "qry_rgb_s5"] = sup_rgb[:, ::5, ::5].numpy(), qry_rgb[:, ::5, ::5].numpy()
        res[e + "sup_fg_bits"] = np.packbits((sup_lab // 255).reshape(-1))
        res[e + "prior_bits"] = np.packbits(prior.numpy().astype(np.uint8).reshape(-1))
        for tag, logits, resp in (("s1_", logits1, resp1), ("s2_", logits2, resp2)):
            pred = logits.argmax(1).numpy().astype(np.uint8)
            res[e + tag + "loss"] = np.array(float(torch.nn.functional.cross_entropy(logits, gt, ignore_index=255)), np.float64)
            res[e + tag + "argmax_bits"] = np.packbits(pred.reshape(-1))
            res[e + tag + "counts"] = _metric_counts(pred[0], qry_lab // 255)
            res[e + tag + "logits_s3"] = logits[0, :, ::3, ::3].numpy()
            res[e + tag + "resp_s3"] = resp[0, ::3, ::3].numpy().astype(np.uint8)
        print("real episode", name, cn, "out", out_shape, "stage-1 loss", float(res[e + "s1_loss"]), "stage-2 loss", float(res[e + "s2_loss"]),
              "fg share", float(gt.float().mean()), "stage-2 fg IoU",
              float(res[e + "s2_counts"][1, 0] / max(res[e + "s2_counts"][1].sum(), 1)))
    np.savez_compressed(OUT / "real_episodes.npz", **res)
    print("wrote real_episodes")


def gen_index_facts():
    """G8: index-map facts of the stock ops."""
    import torch.nn.functional as F
    r = {}
    for (i, o) in ((401, 51), (97, 13)):
        src = F.interpolate(torch.arange(i, dtype=torch.float32).view(1, 1, 1, i), (1, o), mode="nearest")
        r[f"nearest_{i}_{o}"] = src.view(-1).numpy().astype(np.int64)
    r["pool_ceil_201"] = np.array(F.max_pool2d(torch.zeros(1, 1, 201, 201), 3, 2, 1, ceil_mode=True).shape[-1])
    r["pool_ceil_49"] = np.array(F.max_pool2d(torch.zeros(1, 1, 49, 49), 3, 2, 1, ceil_mode=True).shape[-1])
    x = torch.zeros(1, 1, 401, 401)
    sizes = []
    for s in (2, 2, 2, 1):
        x = F.max_pool2d(x, 3, s, 1)
        sizes.append(x.shape[-1])
    r["vgg_sizes_401"] = np.array(sizes)
    np.savez_compressed(OUT / "index_facts.npz", **r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    _install_standins()
    sys.path.insert(0, str(REF))
    small = {"small": ([3, 4], 1, 97, [(97, 97), (80, 120)]),
             "small5": ([5], 5, 97, [(64, 90)])}
    full = {"full": ([5678, 5679], 1, 401, [synth.QUERY_SIZES[5678 % 5], synth.QUERY_SIZES[5679 % 5]])}
    with tempfile.TemporaryDirectory() as tmp:
        only = args.only
        s1 = None
        if only in ("", "stage1", "stage2"):
            s1 = gen_stage1(tmp, "resnet50", "stage1_rn50", {**small, **full} if only != "stage2" else {})
        if only in ("", "map"):
            gen_stage1_map(tmp)
        if only in ("", "stage1rn101"):       # ResNet-101 trunk (layers 3/4/23, networks/pemp_stage1.py:86-96)
            gen_stage1(tmp, "resnet101", "stage1_rn101", {"small": ([3], 1, 97, [(80, 120)])})
        if only in ("", "stage1vgg"):
            gen_stage1(tmp, "vgg16", "stage1_vgg16", {"small": small["small"]})
        if only in ("", "baseline"):
            gen_baseline(tmp, "vgg16", "baseline_vgg16",
                         {"small": small["small"], "small5": small["small5"],
                          "full": ([5678], 1, 401, [synth.QUERY_SIZES[5678 % 5]])})
            gen_baseline(tmp, "resnet50", "baseline_rn50", {"small": small["small"]})
        if only in ("", "stage2"):
            gen_stage2(tmp, s1, {"small": ([3], 1, 97, [(80, 120)]), "small5": ([5], 5, 97, [(64, 90)])})
        if only in ("", "train"):
            gen_train_step(tmp)
        if only in ("", "traj"):
            gen_train_trajectory(tmp)
        if only in ("", "trainfull"):
            gen_train_step(tmp, **TRAIN_FULL)
        if only in ("", "train2full"):
            gen_train_step_stage2(tmp, shot=5, seeds=(1234,), out="stage2_rn50cm_trainstep5_full", H=401)
        if only in ("", "stage2vgg"):
            if s1 is None:
                s1 = gen_stage1(tmp, "resnet50", "stage1_rn50", {})
            gen_stage2_vgg(tmp, s1, {"small": ([3, 4], 1, 97, [(80, 120), (97, 97)]), "small5": ([5], 5, 97, [(64, 90)])})
        if only in ("", "stage2full"):         # BASELINE.json configs[3]: 5-shot stage 2 at 401x401 (12 encoder passes)
            if s1 is None:
                s1 = gen_stage1(tmp, "resnet50", "stage1_rn50", {})
            gen_stage2(tmp, s1, {"full5": ([5678], 5, 401, [synth.QUERY_SIZES[5678 % 5]])})
        if only in ("", "real"):               # the two real PASCAL episodes the reference ships (stage 1 -> prior -> stage 2)
            if s1 is None:
                s1 = gen_stage1(tmp, "resnet50", "stage1_rn50", {})
            gen_real_episodes(tmp, s1)
        if only in ("", "trainbase"):
            gen_train_step_baseline(tmp)
        if only in ("", "trainvgg"):
            gen_train_step_stage1_vgg(tmp)
        if only in ("", "train2"):
            gen_train_step_stage2(tmp)
        if only in ("", "panet"):
            gen_panet(tmp, "vgg16", "panet_vgg16",
                      {"small": small["small"], "small5": small["small5"],
                       "full": ([5678], 1, 401, [synth.QUERY_SIZES[5678 % 5]])})
            gen_panet(tmp, "resnet50", "panet_rn50", {"small": small["small"]})
            gen_train_step_panet(tmp)
        if only in ("", "train5"):
            gen_train_step_5shot(tmp)
            gen_train_step_stage2(tmp, shot=5, seeds=(41, 42), out="stage2_rn50cm_trainstep5")
        if only in ("", "metric"):
            gen_metric()
        if only in ("", "metric"):
            gen_metric_coco()
        if only in ("", "cedt"):
            gen_cedt()
        if only in ("", "facts"):
            gen_index_facts()


if __name__ == "__main__":
    main()
