"""Every tile variant of the training convs (statistics epilogue) on the short-K layers where the autotuner still picks the
64 x 64 tile (profiles/r04_train_picks.txt): python3 scratch/stats_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pemp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (n, hw, cin, cout, k, dil) in ((8, 51, 256, 1024, 1, 1), (8, 51, 128, 512, 1, 1), (8, 51, 256, 512, 1, 1), (8, 101, 64, 256, 1, 1),
                                   (8, 51, 1024, 256, 1, 1), (8, 51, 256, 256, 3, 2)):
    x = torch.randn(n, hw, hw, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) * 0.05
    p = ops.ConvParams(w, None, None, cin, cout, k, k, 1, dil if k == 3 else 0, dil, k * k * cin, False, False)
    m = n * hw * hw
    fl = 2.0 * m * cout * k * k * cin
    res = {}
    for tile in ops._train_tiles(cout):
        try:
            res[tile] = t(lambda: ops.conv2d_stats(x, p, tile=tile))
        except Exception as e:  # noqa: BLE001
            res[tile] = float("nan")
    plain = {tile: t(lambda: ops.conv2d(x, p, tile=tile)) for tile in (23, 24, 25, 26, 27) if cout % ops.TILE_VARIANTS[tile][1] == 0}
    best = min(res, key=lambda k_: res[k_])
    print(f"M {m} {cin}->{cout} k{k}: stats " + "  ".join(f"{k_}:{v:6.1f}" for k_, v in sorted(res.items())) +
          f" | best {best} = {fl / res[best] / 1e6:5.1f} TFLOP/s | plain " + "  ".join(f"{k_}:{v:6.1f}" for k_, v in plain.items()), flush=True)
