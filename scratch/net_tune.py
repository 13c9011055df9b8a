"""In-network tile tuning experiment: after the per-layer autotune, re-pick each layer's tile by the time of the WHOLE
eval step (greedy, one pass over the layers) and report the gain."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from pemp_amd import ops, synth
from pemp_amd.networks import pemp_stage1 as m

dev = torch.device("cuda:0")
net = m.ModelClass(None)
net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
net = net.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
b = synth.make_batch([5678 + i for i in range(B)], shot=1, out_hw=(401, 401))
ins = [torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")]


def step_ms(reps=4):
    with torch.no_grad():
        net.lowres(*ins)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            net.lowres(*ins)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


step_ms(1)                                   # per-layer autotune happens here
base = min(step_ms() for _ in range(3))
print(f"per-layer autotune: {base:.3f} ms/step (eager)")
keys = [k for k in ops._TILE_CACHE if k[-3] == 2 * B]
changed = 0
for k in keys:
    cout = k[1]
    best_t, best = ops._TILE_CACHE[k], min(step_ms() for _ in range(2))
    for t, (bm, bn) in ops.TILE_VARIANTS.items():
        if cout % bn or t == best_t:
            continue
        ops._TILE_CACHE[k] = t
        ms = min(step_ms() for _ in range(2))
        if ms < best * 0.9985:
            best, best_t = ms, t
    if ops._TILE_CACHE[k] != best_t:
        pass
    changed += best_t != ops._TILE_CACHE.get(k) or 0
    ops._TILE_CACHE[k] = best_t
    print(k[:7], "->", best_t, f"{best:.3f}")
final = min(step_ms() for _ in range(3))
print(f"in-network greedy: {final:.3f} ms/step  ({(base / final - 1) * 100:+.2f} %)")
