"""Soak of the one-episode path with split-K variants and four engine lanes: 300 episodes, twice (bit-stable for fixed picks), against
the 25-episode batched path (exact variants): per-episode loss and pixel counts within rounding.  python scratch/soak_lanes.py"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import synth
from pemp_amd.entry import pemp_stage1 as e
dev = torch.device("cuda:0")
net = e.ModelClass(None)
net.load_state_dict(synth.wgen_state_dict_for(net))
net = net.to(dev).eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = e.SyntheticEpisodes(N, 5678, 1, split=0)
data.sample_tasks()
eps = [data.task(i)[:2] for i in range(N)]
ev4 = e.Evaluator(net, device=dev, lanes=4)
t0 = time.time(); r1 = ev4.test_steps_device(eps).cpu(); torch.cuda.synchronize(); t1 = time.time()
r2 = ev4.test_steps_device(eps).cpu()
print("lanes=4 twice identical:", torch.equal(r1, r2), "  %.1f episodes/s incl. first-call capture" % (N / (t1 - t0)))
ev1 = e.Evaluator(net, device=dev, lanes=1)
r3 = ev1.test_steps_device(eps).cpu()
print("lanes=1 vs lanes=4 identical:", torch.equal(r1, r3))
evb = e.Evaluator(net, device=dev)
rows = []
for i in range(0, N, 25):
    rows.append(evb.test_step_batch(eps[i:i + 25]).cpu())
rb = torch.cat(rows)
loss1, lossb = (r1[:, 0] / r1[:, 1]).numpy(), (rb[:, 0] / rb[:, 1]).numpy()
dc = (r1[:, 2:] - rb[:, 2:]).abs().max(dim=1).values.numpy()
print("vs batched exact path: max |d loss| %.2e, episodes with identical counts %d / %d, max count difference %d px, mean %.2f" % (
    np.abs(loss1 - lossb).max(), int((dc == 0).sum()), N, int(dc.max()), dc.mean()))
