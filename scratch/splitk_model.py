import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pemp_amd import ops
dev = torch.device("cuda:0")
def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def run(N, H, ci, co, k, d, tiles=(11, 13, 14, 15, 16, 17)):
    x = torch.randn(N, H, H, ci, device=dev).relu_(); w = torch.randn(co, ci, k, k, device=dev) * 0.05
    pk, kpad = ops.pack_conv_weight(w)
    p = ops.ConvParams(pk, None, None, ci, co, k, k, 1, d * (k // 2), d, kpad, False, False)
    out = ops.conv2d(x, p, tile=13); fl = 2.0 * out.numel() * k * k * ci
    best = min((bench(lambda: ops.conv2d(x, p, out=out, tile=t)), t) for t in tiles if co % {11:128,13:64,14:128,15:64,16:128,17:256}[t] == 0)
    print(f"N={N} H={H} {ci}->{co} k={k}: best tile {best[1]} {best[0]*1e3:.1f} us  {fl/best[0]/1e9:.1f} TF")
    return best[0]
# real layer vs its split-K=3 / split-K=2 equivalents (same blocks x per-block work): M x S rows, K / S
t1 = run(8, 51, 256, 256, 3, 2)
t3 = run(24, 51, 768, 256, 1, 1)     # K=768  = 2304/3, M x3
t2 = run(16, 51, 1152, 256, 1, 1)    # K=1152 = 2304/2, M x2
t4 = run(32, 51, 576, 256, 1, 1)
print(f"split-3 model: {t3/t1:.2f} of the unsplit time; split-2: {t2/t1:.2f}; split-4: {t4/t1:.2f}")
t1 = run(8, 51, 1024, 256, 1, 1); t2 = run(16, 51, 512, 256, 1, 1); t4 = run(32, 51, 256, 256, 1, 1)
print(f"1024->256: split-2 {t2/t1:.2f} split-4 {t4/t1:.2f}")
t1 = run(2, 51, 256, 256, 3, 2); t4 = run(8, 51, 576, 256, 1, 1); t8 = run(16, 51, 288, 256, 1, 1)
print(f"B=1 eval 3x3: split-4 {t4/t1:.2f} split-8 {t8/t1:.2f}")
