"""Baseline VGG-16 train-step gradients: HIP vs the reference's fp32 vs fp64, every sampled tensor (python scratch/vgg_grad_check.py)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import synth
from pemp_amd.networks import baseline as m
from pemp_amd.train_baseline import BaselineTrainer
from tests import util
dev = torch.device("cuda:0")
g, g64 = util.gold("baseline_vgg16_trainstep"), util.gold("baseline_vgg16_trainstep_f64")
net = m.Baseline(None, backbone="vgg16"); net.load_state_dict(util.wgen_state_dict("baseline_vgg16"))
tr = BaselineTrainer(net, device=dev)
b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
t = lambda a: torch.from_numpy(a).to(dev)
loss, _ = tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
print("loss", loss.item(), float(g["loss"]), float(g64["loss64"]))
params = dict(net.named_parameters())
names = [str(n) for n in g["grad_names"]]
for n, r32, r64 in zip(names, g["grad_norms"], g64["grad_norms64"]):
    if r32 < 0: continue
    got = params[n].grad.norm().item()
    print(f"{n:40s} norm hip {got:.7e} ref32 {r32:.7e} f64 {r64:.7e}  rel hip {abs(got-r64)/r64:.1e} ref {abs(r32-r64)/r64:.1e}")
