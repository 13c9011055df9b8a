import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import synth
from pemp_amd.networks import pemp_stage1 as m
from pemp_amd.train_engine import Stage1Trainer
from tests import util
dev = torch.device("cuda:0")
net = m.ModelClass(None); net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
tr = Stage1Trainer(net, device=dev)
b = synth.make_batch([1, 2, 3, 4], shot=1, out_hw=(401, 401))
ins = tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),)
for _ in range(5): tr.train_step(*ins)
torch.cuda.synchronize()
cpu = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); tr.train_step(*ins); cpu.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = (time.perf_counter() - t_all) / 20
print(f"enqueue (CPU) {sum(cpu)/len(cpu)*1e3:.2f} ms/step; wall {tot*1e3:.2f} ms/step")
