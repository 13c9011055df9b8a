"""What-if costing of the training step (timing only, numerics deliberately wrong): how much of the step goes away if
  A  the BatchNorm apply passes of bn1 / bn2 (single-consumer outputs) did not exist (their consumers read z),
  B  DropBlock's pixel_scale passes did not exist (drop_rate 0),
  C  both,
  D  no weight-gradient launch at all: the main chain (forward, input gradients, BatchNorm, head, optimizer) alone,
  E  the weight gradients on the main stream: the fully serialised step.
Usage: python3 scratch/whatif_train.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from pemp_amd import train_engine as te, train_ops as T, ops  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
orig = te.Stage1TrainEngine._cbn_fwd


def no_apply(self, x, conv, bn, relu, residual=None, img_bias=None):
    if not (relu and residual is None and img_bias is None and not conv.stem):
        return orig(self, x, conv, bn, relu, residual, img_bias)
    prm = conv.fwd_params(relu=False, with_bias=False)
    if not ops.stats_supported(x, prm):
        return orig(self, x, conv, bn, relu, residual, img_bias)
    z, part = ops.conv2d_stats(x, prm)
    mean, invstd = bn.stats_from(part, z.numel() // z.shape[-1])
    mask = torch.zeros((z.numel() // z.shape[-1], z.shape[-1] // 32), dtype=torch.int32, device=z.device) if T.mask_supported(z.shape[-1]) else None
    return z, dict(x=x, z=z, y=z, mean=mean, invstd=invstd, relu=relu, mask=mask)


orig_enq = te._enqueue_wgrad


def run(tag, patch, drop, wgrad="side"):
    te.Stage1TrainEngine._cbn_fwd = no_apply if patch else orig
    te._enqueue_wgrad = (lambda *a, **k: None) if wgrad == "none" else orig_enq
    tr = bench.make_trainer("stage1", 1, dev, 0)
    if wgrad == "serial":
        tr.eng.flat.side_stream = tr.eng.buckets.side = None
    if drop is not None:
        tr.eng.drop_rate = drop
    pool = bench.train_pool(dev, 0, 1, 4)
    dt, host, ls, _, _ = bench.timed_train_steps(tr, pool, steps, 5, 1, dev)
    print(f"{tag}: {dt / steps * 1e3:.3f} ms/step", flush=True)
    del tr
    torch.cuda.empty_cache()


run("baseline", False, None)
run("A no bn1/bn2 apply", True, None)
run("B no dropblock", False, 0.0)
run("C both", True, 0.0)
run("D no weight gradients at all (main chain alone)", False, None, wgrad="none")
run("E weight gradients on the main stream (serialised)", False, None, wgrad="serial")
run("baseline again", False, None)
