"""Reads the ISA-level trace left behind the activation tensor by the instrumented kernels (instrument_asm.py)."""
import sys
import numpy as np
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
    t, cb = 1, 0
    kh, kw = divmod(t, 3)
    w = torch.zeros(cout, cin, 3, 3)
    w[0, cb * 32 + 5, kh, kw] = 1.0
    packed, kpad = ops.pack_conv_weight(w.to(dev))
    prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
    y = ops.conv2d(x, prm, pad_value=pv, tile=tile).reshape(-1, cout)
    torch.cuda.synchronize()
    print("row 1300:", float(y[1300, 0]), "(ref 1250)")
    tr = buf[npx + 1:npx + 4].reshape(-1)[:384].view(torch.int32).cpu().numpy().reshape(6, 64)
    names = ["tap", "kw", "kh", "cb", "sa_", "sb_"]
    taph, tapw = dil * W * cin * 4, dil * cin * 4
    print("loop counter (lane) 22 .. 1 = K steps kt0+2 .. kt0+23 of the block; expected sa_ = kh*%d + kw*%d + cb*128, sb_ = tap*%d + cb*128" % (taph, tapw, cin * 4))
    for lane in range(24, 0, -1):
        v = tr[:, lane]
        exp_sa = v[2] * taph + v[1] * tapw + v[3] * 128
        exp_sb = v[0] * cin * 4 + v[3] * 128
        print(f"  ctr {lane:2d}: " + " ".join(f"{n}={int(x_)}" for n, x_ in zip(names, v)) + f"   sa ok {exp_sa == v[4]}  sb ok {exp_sb == v[5]}")

if __name__ == "__main__":
    main()
