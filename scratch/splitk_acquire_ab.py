"""A/B of the split-K hand-off with and without the consumer's agent-scope acquire: run once per library
(PEMP_HIP_LIB=scratch/ab/libpemp_noacq.so = built with -DPEMP_SK_ACQUIRE=0).  Times every split-K tile variant on the layer
shapes that use them: a one-episode evaluation step (M = 5202) and the training step (M = 20808)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def t(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("lib:", os.environ.get("PEMP_HIP_LIB", "default (acquire)"))
for (N, name) in ((2, "eval 1 episode"), (8, "train 4 episodes")):
    for (cin, cout, k, d) in ((256, 256, 3, 2), (1024, 256, 1, 1), (256, 1024, 1, 1), (512, 128, 1, 1)):
        x = torch.randn(N, 51, 51, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        packed, kpad = ops.pack_conv_weight(w)
        prm = ops.ConvParams(packed, None, torch.zeros(cout, device=dev), cin, cout, k, k, 1, d if k == 3 else 0, d, kpad, False, True)
        row = []
        for tile in ops.SPLITK_TILES:
            if cout % ops.TILE_VARIANTS[tile - 10][1]:
                continue
            try:
                us = t(lambda: ops.conv2d(x, prm, tile=tile))
            except Exception as e:                      # noqa: BLE001
                row.append(f"{tile}: {type(e).__name__}")
                continue
            row.append(f"{tile}: {us:6.1f}")
        print(f"{name:17s} M={N * 2601:5d} {cin:4d}->{cout:4d} k{k} d{d} | " + " | ".join(row))
