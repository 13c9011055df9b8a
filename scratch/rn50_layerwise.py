"""Where does a ResNet-50 train-step backward leave the fp64 result?  Gradient at every conv output (dz of the conv+BN pair):
HIP path vs the oracle in fp64 and in fp32 on the CPU.  python scratch/rn50_layerwise.py [baseline|stage1]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from oracle import ref_cpu as R
from pemp_amd import synth
from tests import util
which = sys.argv[1] if len(sys.argv) > 1 else "baseline"
dev = torch.device("cuda:0")
torch.set_num_threads(16)
tag = "baseline_rn50" if which == "baseline" else "stage1_rn50"
sd32 = util.wgen_state_dict(tag)
b = synth.make_batch([31, 32], shot=1, height=97, width=97, out_hw=(97, 97))
tt = lambda a: torch.from_numpy(a)

def cpu_run(dtype):
    sd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd32.items()}
    zs = {}
    orig = R._conv
    def spy(x, sd_, p, *a, **k):
        z = orig(x, sd_, p, *a, **k)
        if z.requires_grad is False:
            z.requires_grad_(True)
        z.retain_grad(); zs[p] = z
        return z
    R._conv = spy; R.TRAIN = True
    try:
        ins = (tt(b["sup_img"]).to(dtype), tt(b["sup_mask"]).to(dtype), tt(b["qry_img"]).to(dtype))
        for k in sd:
            if sd[k].is_floating_point() and "running" not in k:
                sd[k].requires_grad_(True)
        logits = R.baseline_forward(sd, *ins, (97, 97), backbone="resnet50") if which == "baseline" else R.stage1_forward(sd, *ins, (97, 97))
        F.cross_entropy(logits, tt(b["qry_mask"][:, 0]), ignore_index=255).backward()
    finally:
        R._conv = orig; R.TRAIN = False
    return {p: z.grad for p, z in zs.items() if z.grad is not None}

g64, g32 = cpu_run(torch.float64), cpu_run(torch.float32)
if which == "baseline":
    from pemp_amd.networks import baseline as m
    from pemp_amd.train_baseline import BaselineTrainer
    net = m.Baseline(None, backbone="resnet50"); net.load_state_dict(sd32); tr = BaselineTrainer(net, device=dev)
else:
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None); net.load_state_dict(sd32); tr = Stage1Trainer(net, device=dev, drop_rate=0.0)
names = {id(mod): n for n, mod in net.named_modules()}
cap = []
import pemp_amd.train_engine as te
orig_bwd = te.Stage1TrainEngine._cbn_bwd
def spy_bwd(self, dy, rec, conv, bn, **kw):
    out = orig_bwd(self, dy, rec, conv, bn, **kw)
    # the same BatchNorm backward in fp64 on exactly the tensors the kernel saw: kernel error alone
    d, y, z = dy.double(), rec["y"].double(), rec["z"].double()
    c = z.shape[-1]
    g = d * (y > 0) if rec["relu"] else d
    zf, gf = z.reshape(-1, c), g.reshape(-1, c)
    mu, var = zf.mean(0), zf.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    xh = (zf - mu) * inv
    dz64 = (gf - gf.mean(0) - xh * (gf * xh).mean(0)) * (inv * bn.bn.weight.data.double())
    kerr = ((rec["dz"].double().reshape(-1, c) - dz64).norm() / dz64.norm()).item()
    ratio = (mu.abs() / torch.sqrt(var + 1e-5)).max().item()
    cap.append((names[id(conv.conv)], rec["dz"].clone(), kerr, ratio, dy.clone()))
    return out
te.Stage1TrainEngine._cbn_bwd = spy_bwd
t = lambda a: torch.from_numpy(a).to(dev)
tr.forward_backward(t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0]))
torch.cuda.synchronize()
perm = [0, 2, 1, 3]
print("conv (backward order)                         L2rel(hip)  L2rel(cpu32)")
for name, dz, kerr, ratio, dy in cap[:8]:
    ref = g64[name]
    gh = dz.cpu().double().permute(0, 3, 1, 2)[perm]
    rel = lambda a: ((a.double() - ref).norm() / ref.norm()).item()
    print(f"{name:45s} {rel(gh):.2e}    {rel(g32[name]):.2e}   BN-backward kernel vs fp64 on its own inputs {kerr:.2e}   max |mean|/std {ratio:.1f}")
