import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_train_gpu import _trainer, _batch
dev = torch.device("cuda:0")
g = util.gold("stage1_rn50_trainstep")
tr, net = _trainer(dev)
loss, logits = tr.forward_backward(*_batch(dev))
params = dict(net.named_parameters())
for key in [k for k in g.files if k.startswith("grad__")]:
    name = key[len("grad__"):]
    got = params[name].grad.cpu(); ref = torch.from_numpy(g[key])
    got = got if got.numel() <= 40000 else got.reshape(-1)[::37]
    got = got.reshape(ref.shape)
    print(f"{name:50s} maxref {ref.abs().max():.3e} maxerr {(got-ref).abs().max():.3e} rel {((got-ref).abs().max()/ref.abs().max()):.2e} strides {params[name].grad.stride()}")
worst = []
for name, ref in zip(g["grad_names"], g["grad_norms"]):
    if ref > 0: worst.append((abs(params[str(name)].grad.norm().item()-ref)/ref, str(name)))
print(sorted(worst)[-5:])
