"""Second probe of the tile-31 failure (see probe.py): which rows / K steps go wrong.  x[pixel, c] = 1 + pixel + 10000 * (c // 32)
(so the value tells pixel AND channel chunk), one non-zero weight at (tap t, channel c0), pad value = -(1 + c)."""
import sys
import numpy as np
import torch

from pemp_amd import ops


def one(tile, t, c0, dil=1, cin=256, cout=256, N=2, H=51, W=51):
    dev = torch.device("cuda:0")
    npx = N * H * W
    buf = torch.zeros(npx + 4, cin, device=dev)
    buf[:npx, :] = 1.0 + torch.arange(npx, device=dev, dtype=torch.float32)[:, None] + 10000.0 * (torch.arange(cin, device=dev) // 32)[None, :]
    buf[npx, :cin] = -(1.0 + torch.arange(cin, device=dev, dtype=torch.float32))
    x = buf[:npx, :cin].view(N, H, W, cin)
    pv = buf[npx, :cin]
    kh, kw = divmod(t, 3)
    w = torch.zeros(cout, cin, 3, 3)
    w[0, c0, kh, kw] = 1.0
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
    y = ops.conv2d(x, prm, pad_value=pv, tile=tile)
    torch.cuda.synchronize()
    got = y[..., 0].reshape(-1).cpu().numpy()
    ref = ops.conv2d(x, prm, pad_value=pv, tile=27)[..., 0].reshape(-1).cpu().numpy()
    return got, ref


if __name__ == "__main__":
    tile = int(sys.argv[1]) if len(sys.argv) > 1 else 31
    np.set_printoptions(linewidth=250, suppress=True)
    for (t, c0) in ((0, 5), (1, 5), (4, 5), (4, 37), (8, 250)):
        got, ref = one(tile, t, c0)
        bad = got != ref
        print(f"== tile {tile} tap {t} c0 {c0}: {int(bad.sum())} of {bad.size} rows differ")
        # by position of the row in its 128-row tile: which loader wave / DMA instruction / piece
        pos = np.arange(bad.size) % 128
        byw = [(int(bad[(pos % 32) // 8 == wv].sum()), int(((pos % 32) // 8 == wv).sum())) for wv in range(4)]
        byi = [(int(bad[pos // 32 == i].sum()), int((pos // 32 == i).sum())) for i in range(4)]
        print("   bad by loader wave (rows r%32//8):", byw, " by DMA instruction i (row//32):", byi)
        rows = np.arange(256, 256 + 40)
        print("   rows 256..295 got:", got[rows])
        print("   rows 256..295 ref:", ref[rows])
        vals, cnt = np.unique(got[bad] - ref[bad], return_counts=True)
        o = np.argsort(-cnt)[:8]
        print("   most frequent got-ref:", [(float(vals[i]), int(cnt[i])) for i in o])
