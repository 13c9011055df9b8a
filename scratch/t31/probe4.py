"""Debug-dump probe for the instrumented variants (A / D): prints row-1300 value and the 13 debug words the kernel left at
y[tile's first row, n0 + 8 ...] = multi, s_kw, s_ntaps, tap, kh, kw, cb (final), kt0, nkl, a.ntaps, a.KW, s_kw0, s_ntaps0."""
import sys
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
    for (t, cb) in ((0, 0), (1, 0), (4, 0), (3, 5)):
        kh, kw = divmod(t, 3)
        w = torch.zeros(cout, cin, 3, 3)
        w[0, cb * 32 + 5, kh, kw] = 1.0
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
        for rep in range(3):
            y = ops.conv2d(x, prm, pad_value=pv, tile=tile).reshape(-1, cout)
            torch.cuda.synchronize()
            ref = 1 + 1300 + (kh - 1) * dil * W + (kw - 1) * dil + 10000 * cb
            dbg = y[1280, 8:21].cpu().numpy().astype(int).tolist()
            dbg2 = y[1280, 128 + 8:128 + 21].cpu().numpy().astype(int).tolist()
            print(f"t{t} cb{cb} rep{rep}: row1300 {float(y[1300, 0]):.0f} / ref {ref}; dbg(n0=0) {dbg}; dbg(n0=128) {dbg2}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 31)
