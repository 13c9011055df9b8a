"""Un-profiled phase times of the eager training step on the main stream (events at phase boundaries)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pemp_amd import synth
from pemp_amd.train_engine import Stage1Trainer, Stage1TrainEngine
dev = torch.device("cuda:0")
B = 4
pool = []
for g in range(3):
    b = synth.make_batch([1234 + g * B + i for i in range(B)], shot=1, out_hw=(401, 401))
    pool.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),))
net, _ = bench.build_model(None, "stage1", 1)
tr = Stage1Trainer(net, device=dev)
for i in range(10):
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(obj, meth, before, after=None):
    f = getattr(obj, meth)
    def g(*a, **k):
        mark(before)
        r = f(*a, **k)
        if after: mark(after)
        return r
    setattr(obj, meth, g)
eng = tr.eng
wrap(eng, "_trunk_forward", "trunk_fwd")
wrap(eng, "_tail_forward", "tail_fwd", "head")
wrap(eng, "_tail_backward", "tail_bwd")
wrap(eng, "_trunk_backward", "trunk_bwd", "join")
wrap(tr, "optimizer_step", "optimizer", "end")
orig = tr.forward_backward
def fb(*a):
    mark("start")
    return orig(*a)
tr.forward_backward = fb
for i in range(20):
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
import collections
acc = collections.OrderedDict()
for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
    if n0 == "end": continue
    acc.setdefault(n0, []).append(e0.elapsed_time(e1))
tot = 0
for k, v in acc.items():
    print(f"{k:10s} {sum(v)/len(v):7.3f} ms"); tot += sum(v) / len(v)
print("sum", round(tot, 3))
