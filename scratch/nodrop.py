"""Upper bound of what fusing the DropBlock scaling into neighbouring kernels could save: the training step with drop_rate 0."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pemp_amd import synth
from pemp_amd.train_engine import Stage1Trainer
dev = torch.device("cuda:0")
B = 4
pool = []
for g in range(3):
    b = synth.make_batch([1234 + g * B + i for i in range(B)], shot=1, out_hw=(401, 401))
    pool.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),))
for rate in (0.1, 0.0, 0.1, 0.0):
    net, _ = bench.build_model(None, "stage1", 1)
    tr = Stage1Trainer(net, device=dev, drop_rate=rate)
    for i in range(10):
        tr.train_step(*pool[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        tr.train_step(*pool[i % 3])
    torch.cuda.synchronize()
    print("drop_rate", rate, "ms/step", round((time.perf_counter() - t0) / 40 * 1e3, 3), flush=True)
