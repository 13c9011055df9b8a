import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 2 * B
shapes = [  # H, Cin, Cout, k, s, d
    (101, 64, 64, 3, 1, 1), (101, 64, 256, 1, 1, 1), (101, 256, 64, 1, 1, 1),
    (51, 128, 128, 3, 1, 1), (51, 128, 512, 1, 1, 1), (51, 512, 128, 1, 1, 1),
    (51, 256, 256, 3, 1, 2), (51, 256, 1024, 1, 1, 1), (51, 1024, 256, 1, 1, 1), (51, 1024, 512, 1, 1, 1),
]
def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(f"N={N}  shape                       " + "".join(f"tile{t:>2}(TF) " for t in (1, 2, 3)))
for (H, ci, co, k, s, d) in shapes:
    x = torch.randn(N, H, H, ci, device=dev)
    w = torch.randn(co, ci, k, k, device=dev) * 0.05
    pk, kpad = ops.pack_conv_weight(w)
    p = ops.ConvParams(pk, None, None, ci, co, k, k, s, d * (k // 2), d, kpad, False, True)
    out = ops.conv2d(x, p)
    fl = 2.0 * out.numel() * k * k * ci
    row = f"M={N*H*H:6d} N={co:4d} K={k*k*ci:4d}  "
    for t in (1, 2, 3):
        if t == 1 and co % 128: row += "     -     "; continue
        ms = bench(lambda: ops.conv2d(x, p, out=out, tile=t))
        row += f"{fl/ms/1e9:8.1f}   "
    print(row)
