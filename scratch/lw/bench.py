"""Lone-wave K-step cost: the 64x64 tile (id 23) on M = 4096 rows x 256 channels (256 blocks: one per CU, one wave per SIMD) and
M = 8192 (two per SIMD), 3x3 256->256 (72 K steps) and 1x1 1024->256 (32 steps).  Run once per ablated library."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from pemp_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
row = [os.path.basename(os.environ.get("PEMP_HIP_LIB", "default"))]
for (cin, k) in ((256, 3), (1024, 1), (64, 1)):
    for rows_h in (64, 128):
        x = torch.randn(1, rows_h, 64, cin, device=dev)
        w = torch.randn(256, cin, k, k, device=dev) * 0.05
        packed, kpad = ops.pack_conv_weight(w)
        prm = ops.ConvParams(packed, None, None, cin, 256, k, k, 1, k // 2, 1, kpad, False, False)
        us = t(lambda: ops.conv2d(x, prm, tile=23))
        steps = k * k * cin // 32
        row.append(f"{cin}x{k}x{k} M={rows_h * 64}: {us:6.1f} us ({us * 2400 / steps:5.0f} cyc/step)")
print(" | ".join(row))
