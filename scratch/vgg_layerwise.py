"""Where does the Baseline VGG-16 backward leave the fp64 result?  Gradient at every conv output (after the ReLU mask):
HIP path vs the oracle's arithmetic in fp64 (and in fp32 on the CPU).  python scratch/vgg_layerwise.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from oracle import ref_cpu as R
from pemp_amd import synth, train_ops as T
from pemp_amd.networks import baseline as m
from pemp_amd.train_baseline import BaselineTrainer
from tests import util
dev = torch.device("cuda:0")
torch.set_num_threads(16)
sd32 = util.wgen_state_dict("baseline_vgg16")
b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
tt = lambda a: torch.from_numpy(a)

def cpu_run(dtype):
    sd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd32.items()}
    sup, msk, qry = tt(b["sup_img"]).to(dtype), tt(b["sup_mask"]).to(dtype), tt(b["qry_img"]).to(dtype)
    B, S, C, H, W = sup.shape
    x = torch.cat((sup, qry), dim=1).view(B * 2, C, H, W)
    zs = []
    for item in R._VGG:
        if item == "P2": x = F.max_pool2d(x, 3, 2, 1)
        elif item == "P1": x = F.max_pool2d(x, 3, 1, 1)
        else:
            idx, d, relu = item
            z = F.conv2d(x, sd[f"encoder.backbone.features.{idx}.weight"], sd[f"encoder.backbone.features.{idx}.bias"], 1, d, d)
            z.requires_grad_(True); z.retain_grad(); zs.append((idx, z))
            x = F.relu(z) if relu else z
    f = x
    _, c, h, w = f.shape
    f = f.view(B, 2, c, h, w)
    supf = F.interpolate(f[:, :1].reshape(B, c, h, w), (H, W), mode="bilinear", align_corners=True)
    q = f[:, 1:].reshape(B, c, h, w)
    mfg, mbg = msk.view(B, 2, H, W).split(1, dim=1)
    fgv = torch.sum(supf * mfg, dim=(2, 3)) / (mfg.sum(dim=(2, 3)) + 1e-5)
    bgv = torch.sum(supf * mbg, dim=(2, 3)) / (mbg.sum(dim=(2, 3)) + 1e-5)
    pred = R.compute_similarity(fgv.view(B, 1, -1).mean(1), bgv.view(B, 1, -1).mean(1), q, 20)
    logits = F.interpolate(pred, (H, W), mode="bilinear", align_corners=True)
    loss = F.cross_entropy(logits, tt(b["qry_mask"][:, 0]), ignore_index=255)
    loss.backward()
    return {idx: z.grad for idx, z in zs}, {idx: z.detach() for idx, z in zs}

g64, z64 = cpu_run(torch.float64)
g32, z32 = cpu_run(torch.float32)
net = m.Baseline(None, backbone="vgg16"); net.load_state_dict(sd32)
tr = BaselineTrainer(net, device=dev)
cap = []
orig = T.relu_bias_bwd
def spy(dy, y, g, **kw):
    out = orig(dy, y, g, **kw)
    cap.append((g.clone(), y.clone()))
    return out
T.relu_bias_bwd = spy
import pemp_amd.train_baseline as tb
tb.T.relu_bias_bwd = spy
t = lambda a: torch.from_numpy(a).to(dev)
tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
torch.cuda.synchronize()
idxs = [i[0] for i in R._VGG if not isinstance(i, str)]
print("layer    L2rel(hip)  L2rel(cpu32)   sumrel(hip)  sumrel(cpu32)   fwd y L2rel(hip) (cpu32)   mask flips hip / cpu32")
for (g, y), idx in zip(cap, reversed(idxs)):
    perm = [0, 2, 1, 3]          # HIP orders images [all supports | all queries], the reference per episode
    gh = g.cpu().double().permute(0, 3, 1, 2)[perm]
    ref = g64[idx]
    rel = lambda a: ((a.double() - ref).norm() / ref.norm()).item()
    srel = lambda a: (abs(a.double().sum() - ref.sum()) / ref.abs().sum()).item()
    yh = y.cpu().double().permute(0, 3, 1, 2)[perm]
    yref = torch.relu(z64[idx]) if idx != 28 else z64[idx]
    y32 = torch.relu(z32[idx]) if idx != 28 else z32[idx]
    fl_h = int(((yh > 0) != (yref > 0)).sum()); fl_c = int(((y32 > 0) != (yref > 0)).sum())
    print(f"{idx:5d}   {rel(gh):.2e}    {rel(g32[idx]):.2e}      {srel(gh):.2e}    {srel(g32[idx]):.2e}       {((yh - yref).norm() / yref.norm()).item():.2e}  {((y32.double() - yref).norm() / yref.norm()).item():.2e}     {fl_h} / {fl_c}  of {yref.numel()}")
