"""Compact verdict per library variant: (tap, chunk) cases at row 1300 -> value/ref."""
import sys
import torch
from pemp_amd import ops

def main(tile=31, dil=1, cin=256, cout=256, N=2, H=51, W=51):
    dev = torch.device("cuda:0")
    npx = N * H * W
    buf = torch.zeros(npx + 4, cin, device=dev)
    buf[:npx, :] = 1.0 + torch.arange(npx, device=dev, dtype=torch.float32)[:, None] + 10000.0 * (torch.arange(cin, device=dev) // 32)[None, :]
    buf[npx, :cin] = -(1.0 + torch.arange(cin, device=dev, dtype=torch.float32))
    x = buf[:npx, :cin].view(N, H, W, cin)
    pv = buf[npx, :cin]
    res = []
    for (t, cb) in ((0, 0), (1, 0), (3, 0), (4, 0), (8, 0), (0, 3), (5, 3), (3, 5), (7, 7)):
        kh, kw = divmod(t, 3)
        w = torch.zeros(cout, cin, 3, 3)
        w[0, cb * 32 + 5, kh, kw] = 1.0
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
        y = ops.conv2d(x, prm, pad_value=pv, tile=tile).reshape(-1, cout)
        ref = ops.conv2d(x, prm, pad_value=pv, tile=27).reshape(-1, cout)
        torch.cuda.synchronize()
        res.append(f"t{t}c{cb}:{float(y[1300, 0]):.0f}/{float(ref[1300, 0]):.0f}{'' if torch.equal(y, ref) else '!'}")
    print(" ".join(res))

if __name__ == "__main__":
    main()
