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
def head_loss(feat_nhwc, sup_mask, qry_mask, ctr, B, S, Q, protos, dist_scalar, out_shape, weight=None):
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
        pred = torch.stack((bgd, fgd), dim=1).max(dim=2).values
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
