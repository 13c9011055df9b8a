"""Round-5 hunt for the round-4 "tile 31" wrong-result failure: runs against the tree of commit f16640e~1
(exported to scratch/t31/tree, PYTHONPATH points there) -- the build whose padding-value split-K 128x128 4-wave kernel
returned wrong tiles.  Integer probes that tell the candidate mechanisms apart WITHOUT touching the kernel's code:

  x[pixel, c] = pixel index, ONE non-zero weight (tap t, input channel c0, output channel 0), padding value -1000 - c.
  Expected y[m, 0] = pixel(m) + kh*dil*W + kw*dil - pad*(W+1) for in-image taps, else padv[c0].
  Observed displacement d = y - pixel(m) on the rows that disagree:
     d == tap * dil                      -> kw never wraps (s_kw wrong), kh stays 0
     d * ldx*4 == tap*Cin*4 + cb*128     -> the activation DMA ran with the WEIGHTS' scalar offset (whole pixels only if ldx == Cin)
  run with dil in {1, 2, 6} and ldx in {Cin, Cin + 64}.
"""
import sys
import numpy as np
import torch

from pemp_amd import ops


def run(dil, ldx_extra, tile, cin=256, cout=256, N=2, H=51, W=51, reps=6, c0=37):
    dev = torch.device("cuda:0")
    npx = N * H * W
    ldx = cin + ldx_extra
    buf = torch.zeros(npx + 4, ldx, device=dev)
    buf[:npx, :] = torch.arange(npx, device=dev, dtype=torch.float32)[:, None]
    buf[npx, :cin] = -1000.0 - torch.arange(cin, device=dev, dtype=torch.float32)
    x = buf[:npx, :cin].view(N, H, W, cin) if ldx_extra == 0 else buf[:npx].view(N, H, W, ldx)[..., :cin]
    pv = buf[npx, :cin]
    out = {}
    for t in range(9):
        kh, kw = divmod(t, 3)
        w = torch.zeros(cout, cin, 3, 3)
        w[0, c0, kh, kw] = 1.0
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
        # expected
        pix = np.arange(npx).reshape(N, H, W)
        hh = np.arange(H)[None, :, None] + (kh - 1) * dil
        ww = np.arange(W)[None, None, :] + (kw - 1) * dil
        inside = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
        inside = np.broadcast_to(inside, (N, H, W))
        exp = np.where(inside, pix + (kh - 1) * dil * W + (kw - 1) * dil, -1000.0 - c0).astype(np.float32)
        bad_launches, disp_hist, bad_rows_total = 0, {}, 0
        for _ in range(reps):
            y = ops.conv2d(x, prm, pad_value=pv, tile=tile)
            torch.cuda.synchronize()
            got = y[..., 0].cpu().numpy()
            other = float(y[..., 1:].abs().max().item())
            diff = got != exp
            if diff.any() or other != 0.0:
                bad_launches += 1
                bad_rows_total += int(diff.sum())
                d = (got - pix)[diff]
                e = (exp - pix)[diff]
                for a, b in zip(d[:2000].tolist(), e[:2000].tolist()):
                    key = (a if abs(a) < 100000 else "big", b if abs(b) < 100000 else "big")
                    disp_hist[key] = disp_hist.get(key, 0) + 1
        top = sorted(disp_hist.items(), key=lambda kv: -kv[1])[:6]
        out[t] = (bad_launches, bad_rows_total, top)
    return out


if __name__ == "__main__":
    tiles = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [31]
    for tile in tiles:
        for dil in (1, 2, 6):
            for extra in (0, 64):
                try:
                    res = run(dil, extra, tile)
                except Exception as e:                       # noqa: BLE001
                    print(f"tile {tile} dil {dil} ldx+{extra}: {type(e).__name__}: {e}")
                    continue
                nbad = sum(v[0] for v in res.values())
                print(f"tile {tile} dil {dil} ldx+{extra}: bad launches {nbad}")
                for t, (bl, rows, top) in res.items():
                    if bl:
                        print(f"    tap {t}: {bl} bad launches, {rows} bad rows; (observed d, expected d) x count: {top}")
