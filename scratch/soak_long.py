"""Longer soak than soak.py: 1000 synthetic episodes through the batched evaluator (twice: reproducibility), 200 training
steps of stage 1 and 60 of stage 2, all losses finite."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from pemp_amd.entry import pemp_stage1 as e

dev = torch.device("cuda:0")
net = e.ModelClass(None)
net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
net = net.to(dev).eval()
res = []
for rep in range(2):
    ev = e.Evaluator(net, device=dev)
    t0 = time.time()
    res.append(ev.start_eval_loop(e.SyntheticEpisodes(1000, 5678, 1, 0, 401, 401), 20, 0, te_epochs=1, batch=25))
    print(f"eval rep {rep}: loss {res[-1][0]:.6f} mIoU {np.mean(res[-1][1]):.6f}  {1000 / (time.time() - t0):.1f} episodes/s incl. host synthesis", flush=True)
assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]), "eval not reproducible"

from pemp_amd import synth
from pemp_amd.train_engine import Stage1Trainer
from pemp_amd.train_stage2 import Stage2Trainer
from pemp_amd.networks import pemp_stage1 as m1, pemp_stage2 as m2
n1 = m1.ModelClass(None)
n1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
tr = Stage1Trainer(n1, device=dev, lr=1e-3)
torch.manual_seed(0)
pool = []
for g in range(4):
    b = synth.make_batch([100 + 4 * g + i for i in range(4)], shot=1, out_hw=(401, 401))
    pool.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),))
ls = torch.stack([tr.train_step(*pool[i % 4]) for i in range(200)]).cpu().numpy()
assert np.isfinite(ls).all()
print("stage-1 200 steps: loss", ls[:3], "->", ls[-3:], flush=True)
n2 = m2.ModelClass(1, 1, None)
n2.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
tr2 = Stage2Trainer(n1.eval(), n2, device=dev)
ls2 = torch.stack([tr2.train_step(*pool[i % 4]) for i in range(60)]).cpu().numpy()
assert np.isfinite(ls2).all()
print("stage-2 60 steps: loss", ls2[:3], "->", ls2[-3:], flush=True)
print("soak ok")
