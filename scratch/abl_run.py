import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, H, ci, co, k, d) in [(16, 51, 256, 256, 3, 2), (16, 51, 1024, 256, 1, 1), (16, 51, 256, 1024, 1, 1), (64, 51, 256, 256, 3, 2)]:
    x = torch.randn(N, H, H, ci, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.05
    pk, kpad = ops.pack_conv_weight(w)
    p = ops.ConvParams(pk, None, None, ci, co, k, k, 1, d * (k // 2), d, kpad, False, True)
    out = ops.conv2d(x, p); fl = 2.0 * out.numel() * k * k * ci
    r = []
    for t in (11, 12, 13, 14, 15):
        ms = bench(lambda: ops.conv2d(x, p, out=out, tile=t)); r.append(f"t{t}:{fl/ms/1e9:6.1f}")
    print(os.environ.get("PEMP_HIP_LIB", "default")[-8:], f"M={N*H*H} N={co} K={k*k*ci}", " ".join(r))
