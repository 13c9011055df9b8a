import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
N, H, ci, co, k, d = 50, 51, 256, 256, 3, 2
x = torch.randn(N, H, H, ci, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.05
pk, kpad = ops.pack_conv_weight(w)
p = ops.ConvParams(pk, None, None, ci, co, k, k, 1, d * (k // 2), d, kpad, False, True)
out = ops.conv2d(x, p, tile=17)
for t in (17, 14, 16):
    for _ in range(2): ops.conv2d(x, p, out=out, tile=t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d(x, p, out=out, tile=t)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(os.environ.get("PEMP_HIP_LIB", "default")[-12:], "tile", t, f"{ms*1e3:.0f} us {2.0*out.numel()*k*k*ci/ms/1e9:.1f} TF")
