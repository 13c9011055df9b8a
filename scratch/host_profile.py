"""Where the host time of an eager training step goes: cProfile over 10 steps of bench.py's trainer (python scratch/host_profile.py)."""
import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
tr = bench.make_trainer("stage1", 1, dev, 0)
pool = bench.train_pool(dev, 0, 1, 4)
for i in range(4):
    tr.train_step(*pool[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    tr.train_step(*pool[i % 3])
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue ms/step %.2f, wall %.2f" % (host * 100, (time.perf_counter() - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    tr.train_step(*pool[i % 3])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
