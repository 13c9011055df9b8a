import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pemp_amd import ops
dev = torch.device("cuda:0")
NS = int(sys.argv[1]); B = int(sys.argv[2])
nets, pools, streams, wss = [], [], [], []
for s in range(NS):
    net, sd = bench.build_model(dev)
    nets.append(net); pools.append(bench.episode_pool(dev, 1, B, s, n_groups=2)); streams.append(torch.cuda.Stream()); wss.append({})
def step(i):
    for s in range(NS):
        with torch.cuda.stream(streams[s]), torch.no_grad():
            ep = pools[s][i % 2]
            pred, _ = nets[s].lowres_graphed(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
            ops.eval_tail(pred, ep["qry_mask"], ws_cache=wss[s])
for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 30
for i in range(K): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams={NS} batch/stream={B}: {K*NS*B/dt:.1f} episodes/s, {dt/K*1e3:.2f} ms/round")
