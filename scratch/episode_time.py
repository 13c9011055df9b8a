import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pemp_amd.data_kits import synth_u8
from pemp_amd.data_kits.episode import EpisodeTransform, test_samples, train_samples
import random
dev = torch.device("cuda:0")
B = 24
imgs = [(synth_u8.image(k, 375, 500), synth_u8.mask(k, 375, 500)) for k in range(2)]
mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
batch = []
rng = random.Random(0)
for _ in range(B):
    batch += test_samples(imgs[:1], imgs[1:], 401, 401) if mode == "eval" else train_samples(imgs[:1], imgs[1:], 401, 401, rng)
tf = EpisodeTransform(401, 401, device=dev)
t0 = time.perf_counter(); st = tf.stage(batch); t1 = time.perf_counter()
print(f"host stage {1e3*(t1-t0):.2f} ms for {st.nbytes/1e6:.1f} MB")
blob = None
for i in range(3):
    out = tf.run(st, blob); blob = tf._blob
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10):
    out = tf.run(st, blob)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{mode}: upload+preprocess {ms:.3f} ms per batch of {len(batch)} samples; out bytes {out[0].numel()*4/1e6:.1f} MB")
t0 = time.perf_counter()
for i in range(5): st = tf.stage(batch, st.host)
print(f"host stage (reused pinned) {1e3*(time.perf_counter()-t0)/5:.2f} ms")
