"""Soak: N training steps at the full shape (4 episodes, 401x401, DropBlock on) and N eval steps; device memory must not grow,
losses stay finite.  python scratch/soak_train.py [N]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
tr = bench.make_trainer("stage1", 1, dev, 0)
pool = bench.train_pool(dev, 0, 1, 4)
def step(i):
    b = pool[i % len(pool)]
    return tr.train_step(*b)
for i in range(30): step(i)
torch.cuda.synchronize()
m0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
t0 = time.perf_counter(); bad = 0; last = None
for i in range(N):
    l = step(i)
    if i % 100 == 0:
        last = float(l.item()); bad += not (last == last and abs(last) < 1e4)
        print(i, round(last, 4), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"train: {N} steps, {dt / N * 1e3:.2f} ms/step, non-finite {bad}, allocated {m0 >> 20} -> {torch.cuda.memory_allocated() >> 20} MiB, "
      f"reserved {r0 >> 20} -> {torch.cuda.memory_reserved() >> 20} MiB")
