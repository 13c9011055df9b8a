"""Un-profiled: GPU time from the end of the optimizer kernel to the start of the next step's first conv (stem), and from the
last main-stream kernel of the backward pass to the optimizer.  Events only; no profiler."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pemp_amd import _lib, synth
from pemp_amd.train_engine import Stage1Trainer
dev = torch.device("cuda:0")
net, _ = bench.build_model(None, "stage1", 1)
tr = Stage1Trainer(net, device=dev)
B = 4
pool = []
for g in range(3):
    b = synth.make_batch([1234 + g * B + i for i in range(B)], shot=1, out_hw=(401, 401))
    pool.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),))
for i in range(8):
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
real = _lib.load()
marks = []
class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name == "pemp_sgd_clip_step_f32":
            def f(*a):
                rc = fn(*a)
                e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("opt_end", e))
                return rc
            return f
        if name == "pemp_conv2d_nhwc_f32":
            def f(*a):
                if marks and marks[-1][0] == "opt_end":
                    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("first_conv", e))
                return fn(*a)
            return f
        return fn
_lib._lib = Proxy()
import time
t0 = time.perf_counter()
for i in range(30):
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30 * 1e3
_lib._lib = real
gaps = [marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1) if marks[i][0] == "opt_end" and marks[i + 1][0] == "first_conv"]
steps = [marks[i][1].elapsed_time(marks[i + 2][1]) for i in range(len(marks) - 2) if marks[i][0] == "opt_end" and marks[i + 2][0] == "opt_end"]
print("step ms", round(dt, 3), "optimizer end -> first conv of the next step (ms): mean %.3f min %.3f max %.3f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)),
      "n", len(gaps), "opt->opt mean %.3f" % (sum(steps) / max(len(steps), 1)))
