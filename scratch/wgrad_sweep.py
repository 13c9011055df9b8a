"""Weight-gradient kernels alone at the training shapes; PEMP_WGRAD_BLOCKS from the environment (read once per process)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops, train_ops as T
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [  # hw, cin, cout, k, d
    (101, 64, 256, 1, 1), (101, 256, 64, 1, 1), (101, 64, 64, 3, 1), (101, 64, 64, 1, 1),
    (51, 128, 128, 3, 1), (51, 128, 512, 1, 1), (51, 512, 128, 1, 1), (51, 256, 128, 1, 1),
    (51, 256, 256, 3, 2), (51, 256, 1024, 1, 1), (51, 1024, 256, 1, 1), (51, 512, 1024, 1, 1), (51, 1024, 512, 1, 1)]
out = []
for hw, cin, cout, k, d in shapes:
    x = torch.randn(8, hw, hw, cin, device=dev)
    g = torch.randn(8, hw, hw, cout, device=dev)
    prm = ops.ConvParams(None, None, None, cin, cout, k, k, 1, d * (k // 2), d, k * k * cin, False, False)
    dw = torch.empty(cout, k * k * cin, device=dev)
    ws = {}
    us = t(lambda: T.conv_wgrad(x, g, prm, dw, ws_cache=ws, blocks=0))
    fl = 2.0 * 8 * hw * hw * cout * k * k * cin
    out.append(f"{fl/us/1e6:6.1f}")
print(os.environ.get("PEMP_WGRAD_BLOCKS", "768").rjust(5), " ".join(out), flush=True)
