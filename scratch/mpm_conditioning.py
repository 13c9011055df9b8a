# Which rounding of the prototype head dominates the gradient error of a float32 step?  CPU experiment on oracle features:
# the distances of the reference formulation vs the shift-invariant form of csrc/head_common.h.
import torch, sys
import torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from oracle import ref_cpu
from tests import util
from pemp_amd import synth
sd = util.wgen_state_dict("stage1_rn50")
b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
t = lambda a: torch.from_numpy(a)
ref_cpu.TRAIN = True
pred, f = ref_cpu.stage1_forward({k: v.clone() for k, v in sd.items()}, t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), ret_lowres=True)
ref_cpu.TRAIN = False
B, S, Q, p, c = 2, 1, 1, 3, 512
h = w = 13
sup_mask, gt = t(b["sup_mask"]), t(b["qry_mask"][:, 0])
f32 = f.detach()                                            # [B, S+Q, c, h, w]
ctr32 = sd["ctr"]

def head(fe, ctr, dt, dpert=None, win=None, form="direct"):
    fe = fe.to(dt); ctr = ctr.to(dt)
    sup = fe[:, :S].reshape(B * S, c, h * w)
    qry = fe[:, S:].reshape(B * Q, c, 1, h, w)
    m = F.interpolate(sup_mask.view(B * S, 2, 97, 97), (h, w), mode="nearest").to(dt)
    fg, bg = m[:, 0].reshape(B * S, 1, h * w), m[:, 1].reshape(B * S, 1, h * w)
    cc = ctr.view(1, c, 2 * p)
    mask = torch.stack((fg, bg), dim=1)
    if form == "direct":
        D = -((sup.unsqueeze(2) - cc.unsqueeze(3)) ** 2).sum(dim=1)
    else:   # centred, |x|^2 dropped
        cb = cc.view(1, c, 2, p).mean(3, keepdim=True).expand(-1, -1, -1, p).reshape(1, c, 2 * p)
        dc = cc - cb
        bias = (dc * (cc + cb)).sum(1)                       # |c|^2 - |cb|^2
        D = 2 * torch.einsum("bci,cj->bji", sup, dc[0]) - bias[0][None, :, None]
    if dpert is not None:
        D = D + dpert
    Dr = D
    D = (torch.softmax(D.view(-1, 2, p, h * w), dim=2) * mask).view(-1, 1, p * 2, h * w)
    new = ((sup.view(-1, c, 1, h * w) * D).sum(dim=3) / (D.sum(dim=3) + 1e-6)).view(B, S, c, 2, p)
    new = new.transpose(3, 4).reshape(B, S, c * p, 2).mean(dim=1)
    fgp, bgp = new.view(B, c, p, 2).unbind(dim=3)
    fgd = F.cosine_similarity(qry, fgp[..., None, None], dim=1) * 20
    bgd = F.cosine_similarity(qry, bgp[..., None, None], dim=1) * 20
    st = torch.stack((bgd, fgd), dim=1)
    if win is None:
        mv = st.max(dim=2); pred, win = mv.values, mv.indices
    else:
        pred = st.gather(2, win.unsqueeze(2)).squeeze(2)
    logits = F.interpolate(pred, (97, 97), mode="bilinear", align_corners=True)
    return F.cross_entropy(logits, gt, ignore_index=255), win, Dr

def grads(dt, **kw):
    fe = f32.clone().to(dt).requires_grad_(True); ct = ctr32.clone().to(dt).requires_grad_(True)
    loss, win, Dr = head(fe, ct, dt, **kw)
    g = torch.autograd.grad(loss, [fe, ct])
    return loss.item(), [x.double() for x in g], win, Dr.detach()

l64, g64, win, D64 = grads(torch.float64)
def rel(g): return [((a - b).norm() / b.norm()).item() for a, b in zip(g, g64)]
l32, g32, _, D32 = grads(torch.float32, win=win)
print("fp32 everything (direct):", rel(g32), "loss diff", l32 - l64)
# only D's rounding (direct form fp32) injected into the fp64 evaluation
_, gp, _, _ = grads(torch.float64, win=win, dpert=(D32.double() - D64))
print("fp64 + D rounding of fp32 direct:", rel(gp), "max |dD|", (D32.double() - D64).abs().max().item())
# centred form in fp32 end to end
l32c, g32c, _, D32c = grads(torch.float32, win=win, form="centred")
print("fp32 everything (centred):", rel(g32c), "loss diff", l32c - l64)
_, _, _, D64c = grads(torch.float64, win=win, form="centred")
_, gpc, _, _ = grads(torch.float64, win=win, form="centred", dpert=(D32c.double() - D64c))
print("fp64 + D rounding of fp32 centred:", rel(gpc), "max |dD|", (D32c.double() - D64c).abs().max().item())
# fp32 everything but D exact (computed in fp64, cast)
class _: pass
