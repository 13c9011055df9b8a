"""Helpers shared by the parity tests."""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from pemp_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Shape:
    def __init__(self, shape):
        self.shape = tuple(shape)


def key_spec(name):
    with open(os.path.join(GOLD, f"state_keys_{name}.json")) as f:
        return json.load(f)


def wgen_state_dict(name, seed=1234):
    """Wgen weights for the reference key layout `name` as {key: torch tensor}."""
    spec = key_spec(name)
    sd = synth.gen_state_dict({k: _Shape(s) for k, s, _ in spec}, seed)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def perturb_parameters(named_params, seed, rel_eps=1e-7):
    """In place: every parameter tensor of ``named_params`` (an ordered iterable of (name, float tensor): the order of
    Module.named_parameters(), which is the order of the non-buffer keys of state_dict()) multiplied by 1 + rel_eps N(0,1) in
    float64 and rounded back -- about one float32 ulp.  The trajectory fixture's envelope replicas
    (tests/golden/make_golden.py::gen_train_trajectory) and the CPU test that re-creates one of them with the oracle draw
    from the same generator in the same order."""
    gen = torch.Generator().manual_seed(int(seed))
    with torch.no_grad():
        for _, p in named_params:
            p.copy_((p.double() * (1.0 + rel_eps * torch.randn(p.shape, generator=gen, dtype=torch.float64))).to(p.dtype))


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def episode_tensors(seed, shot, H, out_hw, device=None):
    ep = synth.make_episode(int(seed), shot=int(shot), height=int(H), width=int(H), out_hw=tuple(int(v) for v in out_hw))
    t = {k: torch.from_numpy(v)[None] for k, v in ep.items() if k != "cls"}
    t["qry_mask"] = t["qry_mask"][0]          # [1,Ho,Wo]
    if device is not None:
        t = {k: v.to(device) for k, v in t.items()}
    t["cls"] = ep["cls"]
    return t


def counts(pred, ref):
    out = []
    v = ref != 255
    for j in (0, 1):
        out.append([int(((pred == j) & (ref == j) & v).sum()), int(((pred == j) & (ref != j) & v).sum()),
                    int(((pred != j) & (ref == j) & v).sum())])
    return np.array(out, np.int64)


# ---------------------------------------------------------------------------------------------
# prototype head on torch ops (GPU), differentiated by autograd -- reference networks/pemp_stage1.py:142-163,195-261
# (cross-check of the HIP head kernels; test infrastructure)
# ---------------------------------------------------------------------------------------------
def head_loss(feat_nhwc, sup_mask, qry_mask, ctr, B, S, Q, protos, dist_scalar, out_shape, weight=None, win=None):
    """``win`` (int64 [B*Q,2,h,w], channel 0 = background): take these prototypes instead of the group maxima (a frozen
    decision, see test_grad_frozen_gpu.py)."""
    n, h, w, c = feat_nhwc.shape
    f = feat_nhwc.permute(0, 3, 1, 2)
    sup = f[:B * S].reshape(B, S, c, h * w).reshape(B * S, c, h * w)
    qry = f[B * S:].reshape(B * Q, c, 1, h, w)
    H, W = sup_mask.shape[-2:]
    m = F.interpolate(sup_mask.reshape(B * S, 2, H, W), (h, w), mode="nearest")
    fg, bg = m[:, 0].reshape(B * S, 1, h * w), m[:, 1].reshape(B * S, 1, h * w)
    if protos > 0:
        cc = ctr.view(1, c, protos * 2)
        mask = torch.stack((fg, bg), dim=1)
        D = -((sup.unsqueeze(2) - cc.unsqueeze(3)) ** 2).sum(dim=1)
        D = (torch.softmax(D.view(-1, 2, protos, h * w), dim=2) * mask).view(-1, 1, protos * 2, h * w)
        new = ((sup.view(-1, c, 1, h * w) * D).sum(dim=3) / (D.sum(dim=3) + 1e-6)).view(B, S, c, 2, protos)
        new = new.transpose(3, 4).reshape(B, S, c * protos, 2).mean(dim=1)
        fgp, bgp = new.view(B, c, protos, 2).unbind(dim=3)
        fgd = F.cosine_similarity(qry, fgp[..., None, None], dim=1) * dist_scalar
        bgd = F.cosine_similarity(qry, bgp[..., None, None], dim=1) * dist_scalar
        st = torch.stack((bgd, fgd), dim=1)
        pred = st.max(dim=2).values if win is None else st.gather(2, win.unsqueeze(2)).squeeze(2)
    else:
        fgv = (sup * fg).sum(-1) / (fg.sum(-1) + 1e-5)
        bgv = (sup * bg).sum(-1) / (bg.sum(-1) + 1e-5)
        fgp, bgp = fgv.view(B, S, c).mean(1), bgv.view(B, S, c).mean(1)
        q = qry.view(-1, c, h, w)
        pred = torch.stack((F.cosine_similarity(q, bgp[..., None, None], dim=1) * dist_scalar,
                            F.cosine_similarity(q, fgp[..., None, None], dim=1) * dist_scalar), dim=1)
    logits = F.interpolate(pred, out_shape, mode="bilinear", align_corners=True)
    if weight is not None:          # CELossDT: sum(CE * w) / sum(w), core/losses.py:33-43
        ce = F.cross_entropy(logits, qry_mask, ignore_index=255, reduction="none")
        return (ce * weight).sum() / weight.sum(), logits
    return F.cross_entropy(logits, qry_mask, ignore_index=255), logits




def torch_head_step(tr, sup_img, sup_mask, qry_img, qry_msk):
    """Same contract as ``Stage1Trainer.forward_backward`` but with the prototype head evaluated by torch autograd
    (the encoder still runs forward/backward on the HIP engine): returns (loss, full-resolution logits)."""
    eng = tr.eng
    B, S = sup_img.shape[:2]
    Q = qry_img.shape[1]
    eng.flat.attach_grads()
    eng.flat.grad.zero_()
    feat = tr.encode(sup_img, sup_mask, qry_img)
    leaf = feat.detach().requires_grad_(True)
    ctr = tr.model.ctr
    tgt = qry_msk.reshape(-1, *qry_msk.shape[-2:]).contiguous()
    with torch.enable_grad():
        loss, logits = head_loss(leaf, sup_mask, tgt, ctr, B, S, Q, tr.protos, tr.dist_scalar, tuple(qry_msk.shape[-2:]),
                                 weight=tr.loss_obj.weight_map(tgt))
        grads = torch.autograd.grad(loss, [leaf] + ([ctr] if ctr is not None else []))
    eng.flat.attach_grads()
    if ctr is not None:
        ctr.grad.copy_(grads[1])
    eng.backward(grads[0].contiguous())
    return loss.detach(), logits.detach()


# ---------------------------------------------------------------------------------------------
# margin-aware exactness for index outputs (arg-max class, response index): where the winner leads the runner-up
# by more than the float noise between two fp32 summation orders the indices must agree EXACTLY; pixels inside
# the margin are excluded and their fraction reported / bounded
# ---------------------------------------------------------------------------------------------
LOGIT_TOL = 2e-3          # |d logit| bound the end-to-end tests state (logits are 20 * cos, |.| <= 20, i.e. 1e-4 of the
                          # range after ~50 fp32 layers; measured 2e-5 .. 9.3e-4 over the fixtures)
MARGIN = 2 * LOGIT_TOL    # two values that each moved by <= LOGIT_TOL cannot swap order beyond this lead


def assert_argmax_exact(logits, ref_argmax, margin=MARGIN, max_masked=0.03, what=""):
    """logits [B,2,H,W] (ours), ref_argmax [B,H,W] (reference).  Exact agreement outside the margin."""
    logits = logits.detach().cpu()
    ref_argmax = torch.as_tensor(np.asarray(ref_argmax)).long()
    lead = (logits[:, 1] - logits[:, 0]).abs()
    keep = lead > margin
    am = logits.argmax(1)
    wrong = int(((am != ref_argmax) & keep).sum())
    masked = 1.0 - keep.float().mean().item()
    assert wrong == 0, f"{what}: {wrong} arg-max mismatches outside the {margin:g} margin"
    assert masked <= max_masked, f"{what}: {masked:.4f} of the pixels inside the margin"
    return masked


def response_reference(feats_nhwc, sup_mask, ctr, B, S, protos, dist_scalar, out_hw):
    """Response index + its decision margin by the reference's formulas (networks/pemp_stage1.py:195-222,232-261)
    evaluated on the given feature map (fixture tests pass the ORACLE's, ``oracle_stage1_feats``; the kernel tests of
    test_ops_gpu.py their own random operands): returns (response int64
    [B,Ho,Wo], margin f32 [B,Ho,Wo]) at the output size (nearest upsample, pemp_stage1.py:162).  margin = the
    smaller of: lead of the winning prototype inside the winning class, lead of the winning class."""
    from oracle import ref_cpu
    f = feats_nhwc.detach().cpu().permute(0, 3, 1, 2).contiguous()
    n, c, h, w = f.shape
    Q = n // B - S
    sup_f, qry_f = f[:B * S].view(B, S, c, h, w), f[B * S:].view(B, Q, c, h, w)     # images are [all supports | all queries]
    H, W = sup_mask.shape[-2:]
    m = F.interpolate(sup_mask.detach().cpu().reshape(B * S, 2, H, W), (h, w), mode="nearest")
    fg, bg = m.unbind(dim=1)
    _, _, ap = ref_cpu.mpm(sup_f, qry_f, fg, bg, ctr.detach().cpu(), protos, dist_scalar, True)
    fgp, bgp = ap.view(B, c, 2, protos).unbind(dim=2)                      # adaptive_p = [B, c, (fg|bg) x p]
    sim = ref_cpu.compute_similarity(fgp, bgp, qry_f.reshape(-1, c, 1, h, w), dist_scalar)     # [B,2,p,h,w] (bg, fg)
    top = sim.topk(2, dim=2)
    vals, idx = top.values, top.indices
    cls = (vals[:, 1, 0] > vals[:, 0, 0]).long()                            # ties -> class 0, as argmax does
    inner = vals[:, :, 0] - vals[:, :, 1]                                   # [B,2,h,w]
    pick = lambda t: torch.where(cls == 1, t[:, 1], t[:, 0])
    resp = pick(idx[:, :, 0]) + 3 * cls
    margin = torch.minimum(pick(inner), (vals[:, 1, 0] - vals[:, 0, 0]).abs())
    up = lambda t: F.interpolate(t[:, None].float(), tuple(out_hw), mode="nearest")[:, 0]
    return up(resp).long(), up(margin)


def oracle_stage1_feats(sd, sup_img, qry_img, backbone="resnet50"):
    """The ORACLE's encoder output for an episode batch, laid out as the engines lay theirs out (NHWC, [all supports | all
    queries]) -- what ``response_reference`` takes: the decision margins of a fixture test then come from the reference's
    arithmetic on the CPU, not from the features of the model under test."""
    from oracle import ref_cpu
    sup_img, qry_img = sup_img.detach().cpu(), qry_img.detach().cpu()
    B, S, ch, H, W = sup_img.shape
    Q = qry_img.shape[1]
    with torch.no_grad():
        f = ref_cpu.encoder_stage1(torch.cat((sup_img, qry_img), dim=1).view(B * (S + Q), ch, H, W), sd, backbone)
    f = f.view(B, S + Q, *f.shape[1:])
    return torch.cat((f[:, :S].flatten(0, 1), f[:, S:].flatten(0, 1))).permute(0, 2, 3, 1).contiguous()


def oracle_stage2_feats(sd, sup_img, sup_mask, qry_img, qry_prior, backbone2="resnet50"):
    """As ``oracle_stage1_feats`` for PEMPStage2's encoder (images + prior channel, communication modules)."""
    from oracle import ref_cpu
    sup_img, sup_mask, qry_img, qry_prior = (t.detach().cpu() for t in (sup_img, sup_mask, qry_img, qry_prior))
    B, S = sup_img.shape[:2]
    Q = qry_img.shape[1]
    with torch.no_grad():
        f = ref_cpu.encoder_stage2(sd, sup_img, sup_mask, qry_img, qry_prior, backbone2)
    f = f.view(B, S + Q, *f.shape[1:])
    return torch.cat((f[:, :S].flatten(0, 1), f[:, S:].flatten(0, 1))).permute(0, 2, 3, 1).contiguous()


def assert_response_exact(resp_got, resp_ref, margin_map, margin=MARGIN, what="", max_masked=0.12):
    """Exact agreement of the response index wherever its decision margin exceeds ``margin``.  Returns the masked
    fraction (coinciding meta-prototypes -- centres that attract no pixel pool to the same vector -- give exact ties
    over whole regions; those are the masked pixels).  ``max_masked``: measured share + 0.05 -- every fixture but one has at
    most 6.0 % of its pixels inside the margin (default bound 0.12); stage1_rn50_small's first episode has 12.2 % (its caller
    passes 0.18)."""
    resp_got = torch.as_tensor(np.asarray(resp_got.cpu() if hasattr(resp_got, "cpu") else resp_got)).long()
    resp_ref = torch.as_tensor(np.asarray(resp_ref)).long()
    keep = margin_map > margin
    wrong = int(((resp_got != resp_ref) & keep).sum())
    masked = 1.0 - keep.float().mean().item()
    print(f"response index {what}: {masked:.4f} of the pixels inside the {margin:g} margin (bound {max_masked:.4f}), {wrong} mismatches outside")
    assert wrong == 0, f"{what}: {wrong} response-index mismatches outside the {margin:g} margin"
    assert masked <= max_masked, f"{what}: {masked:.4f} of the response pixels inside the margin"
    return masked


# ---------------------------------------------------------------------------------------------
# gradients of a training step against the reference's fp32 gradients AND an fp64 evaluation of the same step
# ---------------------------------------------------------------------------------------------
def check_gradients(g, g64, params, what, eps=3e-3):
    """Gradients of one training step: ``g`` = the reference's own fp32 gradients (fixture made by the reference),
    ``g64`` = the same step evaluated in fp64 by the oracle under autograd, ``params`` = the HIP path's.
    The step is not a smooth function of its rounding -- the prototype max (compute_similarity(...).max(dim=2)), max-pooling
    and every ReLU switch discretely, so a 1e-7 change of a batch statistic moves single pixels between branches and single
    gradient entries by ~1e-3 of the tensor's maximum (seen when only the partial-sum order of the BatchNorm reductions
    changed); the reference's own fp32 error scatters between 1e-6 and 6e-3 of max|g| from tensor to tensor.  So the bound is
    RELATIVE to that error, per tensor, in two norms (factor 3: the reference's fp32 run shares its tie-breaking and op
    structure with the fp64 evaluation, so its switches coincide with fp64's more often than another implementation's do):
        L2 :  |hip - g64|_2 <= 3 * |ref32 - g64|_2 + eps * |g64|_2
        max:  |hip - g64|_oo <= 3 * |ref32 - g64|_oo + eps * |g64|_oo
    and for every parameter's gradient NORM  |n_hip - n_64| <= 2 * |n_ref32 - n_64| + eps * n_64.
    ``eps`` = 3e-3: one switched arg-max of VGG-16's stride-1 max pool already costs 1e-3 in L2 (scratch/vgg_layerwise.py: the
    gradient is exact to 3e-6 above that pool and 1.3e-3 off below it, with no ReLU mask differing)."""
    bad, worst = [], 0.0
    for name, ref, ref64 in zip(g["grad_names"], g["grad_norms"], g64["grad_norms64"]):
        p = params[str(name)]
        if ref < 0:
            assert not p.requires_grad, name
            continue
        assert p.requires_grad, name
        got = p.grad.norm().item()
        if abs(got - ref64) > 2 * abs(ref - ref64) + eps * ref64 + 1e-6:
            bad.append((str(name), got, float(ref), float(ref64)))
    assert not bad, (what, bad[:10])
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        got = params[name].grad.cpu()
        ref = torch.from_numpy(g[key])
        got = (got if got.numel() <= 40000 else got.reshape(-1)[::37]).reshape(ref.shape)
        ref64 = torch.from_numpy(g64["g64__" + name])
        scale, n2 = ref64.abs().max().item(), ref64.norm().item()
        e_ref, e_hip = (ref.double() - ref64).abs().max().item(), (got.double() - ref64).abs().max().item()
        l_ref, l_hip = (ref.double() - ref64).norm().item(), (got.double() - ref64).norm().item()
        print(f"{what} {name:46s} max: hip {e_hip / max(scale, 1e-30):.1e} ref32 {e_ref / max(scale, 1e-30):.1e} | "
              f"L2: hip {l_hip / max(n2, 1e-30):.1e} ref32 {l_ref / max(n2, 1e-30):.1e}")
        assert l_hip <= 3 * l_ref + eps * n2 + 1e-7, (what, name, "L2", l_hip / max(n2, 1e-30), l_ref / max(n2, 1e-30))
        assert e_hip <= 3 * e_ref + eps * scale + 1e-7, (what, name, "max", e_hip / max(scale, 1e-30), e_ref / max(scale, 1e-30))
        worst = max(worst, e_hip / (3 * e_ref + eps * scale + 1e-7))
    return worst


def check_gradients_live(hip, g32, g64, what, eps=3e-3):
    """check_gradients for gradients computed in the test itself: ``hip`` / ``g32`` / ``g64`` = {name: tensor} of the HIP path,
    of the oracle in float32 and of the oracle in float64 on the same step, WHOLE tensors.  The L2 and the norm bound are those
    of check_gradients; its max-norm bound is not applied: one ReLU that switches at one of the 676 pixels of a 97 x 97 step
    moves a whole row of a 1 x 1 conv's weight gradient, measured up to 1.7e-2 of the tensor's maximum with nothing wrong in
    L2 (2e-3, the switching floor of an unfrozen step) -- on whole tensors (the fixture check samples every 37th element) the maximum finds such an entry every run."""
    assert set(hip) == set(g64) == set(g32), sorted(set(hip) ^ set(g64))[:10]
    rows = []
    for name, r64 in g64.items():
        got, r32 = hip[name].double().cpu(), g32[name].double()
        n2 = r64.norm().item()
        l_ref, l_hip = (r32 - r64).norm().item(), (got - r64).norm().item()
        if n2 > 1e-12:                                            # exact zeros (see test_stage2_train_step_matches_reference) are held by the absolute terms only
            rows.append((l_hip / n2, l_ref / n2, name))
        assert l_hip <= 3 * l_ref + eps * n2 + 1e-7, (what, name, "L2", l_hip / max(n2, 1e-30), l_ref / max(n2, 1e-30))
        assert abs(got.norm().item() - n2) <= 2 * abs(r32.norm().item() - n2) + eps * n2 + 1e-6, (what, name, "norm")
    rows.sort(reverse=True)
    print(f"{what}: {len(rows)} tensors, relative L2 error vs float64: hip max {rows[0][0]:.2e} ({rows[0][2]}), "
          f"median {rows[len(rows) // 2][0]:.2e}; oracle float32 max {max(r[1] for r in rows):.2e}")
    return rows


# ---------------------------------------------------------------------------------------------
# a tiny PASCAL-5i directory tree (reference layout, data_kits/pascal_voc.py:103-107,262-264) written from synthetic pictures
# ---------------------------------------------------------------------------------------------
def make_tiny_voc(root, per_class=(4, 5, 6, 7), splits=("train", "val"), seed=0):
    """<root>/JPEGImages/<name>.jpg, <root>/Binary_map_aug/{train,val}/<cls>.txt and .../<cls>/<name>.png for the 20 classes:
    class c has per_class[c % len(per_class)] samples named like VOC's ("2007_00c0k"), pictures of four sizes from
    pemp_amd.data_kits.synth_u8 (JPEG quality 95), 0 / 255 label images.  -> {(mode, cls): [names]}."""
    from pathlib import Path
    from PIL import Image
    from pemp_amd.data_kits import synth_u8
    root = Path(root)
    (root / "JPEGImages").mkdir(parents=True, exist_ok=True)
    sizes = ((120, 160), (150, 130), (133, 177), (160, 160))
    lists = {}
    for mode in splits:
        d = root / "Binary_map_aug" / mode
        d.mkdir(parents=True, exist_ok=True)
        for c in range(1, 21):
            names = [f"20{7 + (c + k) % 6:02d}_{'0' if mode == 'val' else '1'}{c:02d}{k:02d}" for k in range(per_class[c % len(per_class)])]
            (d / f"{c}.txt").write_text("\n".join(names) + "\n")
            (d / str(c)).mkdir(exist_ok=True)
            for k, n in enumerate(names):
                h, w = sizes[(c + k) % len(sizes)]
                s = seed + 1000 * c + 10 * k + (1 if mode == "val" else 0)
                if not (root / "JPEGImages" / f"{n}.jpg").exists():
                    Image.fromarray(synth_u8.image(s, h, w)).save(root / "JPEGImages" / f"{n}.jpg", quality=95)
                Image.fromarray(synth_u8.mask(s, h, w)).save(d / str(c) / f"{n}.png")
            lists[(mode, c)] = names
    return lists
