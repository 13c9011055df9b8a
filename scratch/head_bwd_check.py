"""dL/dfeat of the HIP prototype head vs torch autograd in fp64 on the features the HIP encoder produced:
python scratch/head_bwd_check.py   (stage-1 VGG-16 and ResNet-50, fixture batch)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import synth, ops, train_ops as T
from pemp_amd.networks import pemp_stage1 as m
from pemp_amd.train_engine import Stage1Trainer
from tests import util
dev = torch.device("cuda:0")
b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
t = lambda a: torch.from_numpy(a).to(dev)
sup, msk, qry, gt = t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])
for bb, tag in (("vgg16", "stage1_vgg16"), ("resnet50", "stage1_rn50")):
    net = m.ModelClass(None, backbone=bb)
    net.load_state_dict(util.wgen_state_dict(tag))
    tr = Stage1Trainer(net, device=dev, drop_rate=0.0)
    grabbed = {}
    orig = tr.eng.backward
    tr.eng.backward = lambda dfeat: grabbed.setdefault("dfeat", dfeat.clone())
    tr.eng.flat.attach_grads(); tr.eng.flat.grad.zero_()
    feat = tr.encode(sup, msk, qry)
    loss, pred = tr._head_hip(feat, msk, gt, 2, 1, 1)
    ctr_g = net.ctr.grad.clone()
    leaf = feat.detach().double().requires_grad_(True)
    ctr = net.ctr.detach().double().requires_grad_(True)
    l64, _ = util.head_loss(leaf, msk.double(), gt, ctr, 2, 1, 1, tr.protos, tr.dist_scalar, (97, 97))
    g64, c64 = torch.autograd.grad(l64, [leaf, ctr])
    leaf32 = feat.detach().clone().requires_grad_(True)
    ctr32 = net.ctr.detach().clone().requires_grad_(True)
    l32, _ = util.head_loss(leaf32, msk, gt, ctr32, 2, 1, 1, tr.protos, tr.dist_scalar, (97, 97))
    g32, c32 = torch.autograd.grad(l32, [leaf32, ctr32])
    rel = lambda a, r: ((a.double() - r).norm() / r.norm()).item()
    print(f"{tag}: loss hip {loss.item():.7f} f64 {l64.item():.7f} | dfeat L2 rel: hip {rel(grabbed['dfeat'], g64):.2e} torch32 {rel(g32, g64):.2e} | "
          f"dctr: hip {rel(ctr_g, c64):.2e} torch32 {rel(c32, c64):.2e} | colsum(dfeat) rel: hip {rel(grabbed['dfeat'].sum((0,1,2)), g64.sum((0,1,2))):.2e} "
          f"torch32 {rel(g32.sum((0,1,2)), g64.sum((0,1,2))):.2e} | |feat| mean {feat.abs().mean().item():.3f}")
