import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from oracle import ref_cpu
from pemp_amd.networks import pemp_stage1 as m
import torch.nn.functional as F
dev = torch.device("cuda:0")
sd = util.wgen_state_dict("stage1_rn50")
net = m.ModelClass(None); net.load_state_dict(sd); net = net.to(dev).eval()
t = util.episode_tensors(3, 1, 97, (97, 97))
with torch.no_grad():
    pred, resp = net.lowres(t["sup_img"].to(dev), t["sup_mask"].to(dev), t["qry_img"].to(dev), ret_ind=True)
    # oracle lowres
    B,S,ch,H,W = t["sup_img"].shape
    x = torch.cat((t["sup_img"], t["qry_img"]), 1).view(2, ch, H, W)
    f = ref_cpu.encoder_stage1(x, sd)
    _, c, h, w = f.shape
    f = f.view(1, 2, c, h, w)
    mk = F.interpolate(t["sup_mask"].view(1, 2, H, W), (h, w), mode="nearest")
    rp, rr, ap = ref_cpu.mpm(f[:, :1], f[:, 1:], mk[:, 0], mk[:, 1], sd["ctr"], 3, 20, True)
print("pred diff", (pred.cpu() - rp).abs().max().item())
print("resp agree", (resp.cpu().long() == rr).float().mean().item())
print("mine", resp.cpu()[0, :4])
print("ref ", rr[0, :4])
pro = net._last_protos.cpu()   # [1,6,c]
print("protos diff", (pro.permute(0, 2, 1) - ap).abs().max().item(), ap.abs().max().item())
# cos of each proto
q = f[0, 1]  # c,h,w
for j in range(6):
    cs = F.cosine_similarity(q[None], ap[0, :, j][None, :, None, None], dim=1) * 20
    print(j, cs[0, 0, :5])
