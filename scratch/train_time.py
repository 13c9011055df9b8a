import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from pemp_amd import synth
from pemp_amd.networks import pemp_stage1 as m
from pemp_amd.train_engine import Stage1Trainer
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = m.ModelClass(None); net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
tr = Stage1Trainer(net, device=dev)
b = synth.make_batch(list(range(100, 100 + B)), shot=1, out_hw=(401, 401))
t = lambda a: torch.from_numpy(a).to(dev)
sup, msk, qry, gt = t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])
for _ in range(2): tr.train_step(sup, msk, qry, gt)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): loss = tr.train_step(sup, msk, qry, gt)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={B} train step {dt*1e3:.2f} ms -> {B/dt:.1f} episodes/s, loss {loss.item():.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
