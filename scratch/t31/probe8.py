"""Verdict + the six words an edit_asm.py --dump build left behind the activation tensor."""
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
    for (t, cb) in ((1, 0), (4, 0), (7, 7)):
        kh, kw = divmod(t, 3)
        w = torch.zeros(cout, cin, 3, 3)
        w[0, cb * 32 + 5, kh, kw] = 1.0
        packed, kpad = ops.pack_conv_weight(w.to(dev))
        prm = ops.ConvParams(packed, None, None, cin, cout, 3, 3, 1, dil, dil, kpad, False, False)
        buf[npx + 1:] = 0
        y = ops.conv2d(x, prm, pad_value=pv, tile=tile).reshape(-1, cout)
        torch.cuda.synchronize()
        tr = buf[npx + 1:npx + 4].reshape(-1)[:384].view(torch.int32).cpu().numpy().reshape(6, 64)
        ref = ops.conv2d(x, prm, pad_value=pv, tile=27).reshape(-1, cout)
        res.append(f"t{t}c{cb}:{float(y[1300, 0]):.0f}/{float(ref[1300, 0]):.0f}{'' if torch.equal(y, ref) else '!'} dump(lane0)={tr[:, 0].tolist()} lanes equal={bool((tr == tr[:, :1]).all())}")
    print("\n".join(res))

if __name__ == "__main__":
    main()
