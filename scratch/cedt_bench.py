"""CELossDT weight map on the device (pemp_cedt_weight_f32) at the bench shapes: 25 ground truths of one PASCAL format and 4 of
401 x 401 (the training step's).  Run under rocprofv3 --kernel-trace --stats for the per-kernel split.
    python3 scratch/cedt_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pemp_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
for B, hw in ((25, (375, 500)), (25, (500, 333)), (4, (401, 401))):
    b = synth.make_batch([5678 + i for i in range(min(B, 5))], shot=1, out_hw=hw)
    t = torch.from_numpy(b["qry_mask"][:, 0]).to(dev)
    t = t.repeat((B + t.shape[0] - 1) // t.shape[0], 1, 1)[:B].contiguous()
    ws = {}
    for _ in range(3):
        ops.cedt_weight(t, 5.0, ws_cache=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        w = ops.cedt_weight(t, 5.0, ws_cache=ws)
    e1.record()
    torch.cuda.synchronize()
    print(f"cedt_weight B={B} {hw}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call; checksum {float(w.double().sum()):.6f}", flush=True)
