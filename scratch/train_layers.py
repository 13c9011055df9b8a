import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, train_ops as T, synth
from pemp_amd.networks import pemp_stage1 as m
from pemp_amd.train_engine import Stage1Trainer
from pemp_amd import synth as _synth
dev = torch.device("cuda:0")
net = m.ModelClass(None); net.load_state_dict(_synth.wgen_state_dict_for(net))
tr = Stage1Trainer(net, device=dev)
b = synth.make_batch([1, 2, 3, 4], shot=1, out_hw=(401, 401))
ins = tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) + (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),)
for _ in range(3): tr.train_step(*ins)
rec = []
def wrap(mod, name, keyfn):
    orig = getattr(mod, name)
    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = orig(*a, **k); e1.record()
        rec.append((name,) + keyfn(a, k, y) + (e0, e1))
        return y
    setattr(mod, name, timed)
def conv_key(a, k, y):
    x, p = a[0], a[1]; n, ho, wo, co = y.shape
    return ((n*ho*wo, co, p.kh*p.kw*p.cin, p.kh, p.stride), 2.0*n*ho*wo*co*p.kh*p.kw*(3 if p.stem else p.cin))
def wg_key(a, k, y):
    x, g, p, dw = a[:4]; m_ = g.numel() // p.cout
    return ((m_, p.cout, p.kh*p.kw*p.cin, p.kh, p.stride), 2.0*m_*p.cout*p.kh*p.kw*(3 if p.stem else p.cin))
wrap(ops, "conv2d", conv_key); wrap(T, "conv_wgrad", wg_key)
import pemp_amd.train_engine as te
for i in range(3):
    if i == 1: rec.clear()
    tr.train_step(*ins)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, fl, e0, e1 in rec:
    a = agg.setdefault((name, key), [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
for which in ("conv2d", "conv_wgrad"):
    tot = sum(v[1] for k, v in agg.items() if k[0] == which) / 2
    print(which, f"total {tot:.2f} ms/step")
    for (name, key), (cnt, ms, fl) in agg.items():
        if name != which: continue
        print(f"  M={key[0]:7d} N={key[1]:5d} K={key[2]:5d} k={key[3]} s={key[4]} cnt={cnt//2:3d} {ms/cnt*1e3:8.1f} us {fl/ms/1e9:7.1f} TF {ms/2/tot*100:5.1f}%")
