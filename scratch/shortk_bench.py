"""Short-K layers of the training step (M = 20 808): plain conv, conv + BatchNorm statistics, input gradient + BatchNorm backward
sums, per tile variant, alone on the chip (hipGraph of 10 launches).  python scratch/shortk_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pemp_amd import ops
dev = torch.device("cuda:0")
N, H, W = 8, 51, 51
M = N * H * W


def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for cin, cout in ((256, 1024), (128, 512), (1024, 256), (512, 128)):
    x = torch.randn(N, H, W, cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    pk, kp = ops.pack_conv_weight(w)
    p = ops.ConvParams(pk, None, None, cin, cout, 1, 1, 1, 0, 1, kp, False, False)
    z = torch.randn(N, H, W, cout, device=dev)
    res = torch.randn(N, H, W, cout, device=dev)
    out = torch.empty(N, H, W, cout, device=dev)
    mask = torch.randint(-2**31, 2**31 - 1, (M, cout // 32), dtype=torch.int32, device=dev)
    bn = {"z": z, "mask": mask, "mean": torch.randn(cout, device=dev), "invstd": torch.rand(cout, device=dev) + 0.5}
    gf = 2.0 * M * cin * cout / 1e9
    print("1x1 %d -> %d at M = %d (%.1f GFLOP; output %.0f MB)" % (cin, cout, M, gf, M * cout * 4 / 1e6))
    for tile in (23, 22, 25, 21, 24, 31, 34, 35):
        row = []
        for name, fn in (("conv", lambda: ops.conv2d(x, p, out=out, tile=tile)),
                         ("conv+res", lambda: ops.conv2d(x, p, out=out, residual=res, tile=tile)),
                         ("stats", lambda: ops.conv2d_stats(x, p, out=out, tile=tile)),
                         ("bnbwd", lambda: ops.conv2d_bnbwd(x, p, bn, out=out, tile=tile)),
                         ("bnbwd+res", lambda: ops.conv2d_bnbwd(x, p, bn, residual=res, out=out, tile=tile))):
            try:
                us = timeit(fn)
                row.append("%s %6.1f us %5.1f TF" % (name, us, gf / us * 1e3))
            except Exception as e:  # noqa: BLE001
                row.append("%s  n/a" % name)
        print("   tile %2d: " % tile + " | ".join(row), flush=True)
