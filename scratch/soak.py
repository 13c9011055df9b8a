"""Longer runs than the tests: determinism of the graphed eval path, loader/stream interplay, 60 training steps."""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from pemp_amd.entry import pemp_stage1 as e
from pemp_amd.entry import train_stage1 as t
dev = torch.device("cuda:0")
net = e.ModelClass(None); net.load_state_dict(util.wgen_state_dict("stage1_rn50")); net = net.to(dev).eval()
res = []
for rep in range(2):
    ev = e.Evaluator(net, device=dev)
    t0 = time.time()
    res.append(ev.start_eval_loop(e.SyntheticDecodedEpisodes(150, 5678, 1, split=0), 20, 0, te_epochs=2))
    print(f"eval rep {rep}: loss {res[-1][0]:.6f} mIoU {np.mean(res[-1][1]):.6f}  {300/(time.time()-t0):.1f} episodes/s (batch 1, incl. host synth)")
assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]), "eval not reproducible"
torch.manual_seed(0)
model = t.main(steps=60, bs=4, shot=1, lr=1e-3, seed=3, log_every=20, model="stage1", decoded=1, height=201, width=201)
print("train ok", all(torch.isfinite(p).all().item() for p in model.parameters()))
