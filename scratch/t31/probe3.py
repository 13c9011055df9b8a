"""Third probe: the full (tap, channel chunk) map.  One non-zero weight per launch at (tap t, channel 32 cb + 5); x[pixel, c] =
1 + pixel + 10000 (c // 32); the value at an interior row names the pixel displacement and the chunk of every contribution."""
import sys
import numpy as np
import torch
from pemp_amd import ops

def main(tile, dil=1, cin=256, cout=256, N=2, H=51, W=51):
    dev = torch.device("cuda:0")
    npx = N * H * W
    buf = torch.zeros(npx + 4, cin, device=dev)
    buf[:npx, :] = 1.0 + torch.arange(npx, device=dev, dtype=torch.float32)[:, None] + 10000.0 * (torch.arange(cin, device=dev) // 32)[None, :]
    buf[npx, :cin] = -(1.0 + torch.arange(cin, device=dev, dtype=torch.float32))
    x = buf[:npx, :cin].view(N, H, W, cin)
    pv = buf[npx, :cin]
    rows = [1300, 1301, 1300 + 2601 + 7]
    print("rows", rows, "(pixel index = row); ref value = 1 + row + (kh-1)*dil*W + (kw-1)*dil + 10000*cb")
    for cb in range(cin // 32):
        line = []
        for t in range(9):
            kh, kw = divmod(t, 3)
            w = torch.zeros(cout, cin, 3, 3)
            w[0, cb * 32 + 5, kh, kw] = 1.0
            packed, kpad = ops.pack_conv_weight(w.to(dev))
            prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
            y = ops.conv2d(x, prm, pad_value=pv, tile=tile)[..., 0].reshape(-1)
            torch.cuda.synchronize()
            ref = 1 + rows[0] + (kh - 1) * dil * W + (kw - 1) * dil + 10000 * cb
            g = [float(y[r]) for r in rows]
            line.append(f"t{t}: {g[0]:.0f}/{ref} ({g[0] / ref:.3f}; r1-r0 {g[1] - g[0]:.0f}, r2-r0 {g[2] - g[0]:.0f})")
        print(f"cb {cb}: " + " | ".join(line))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 31)
