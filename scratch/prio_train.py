"""Does HIP stream priority help the training step?  Main chain on a high-priority stream (weight gradients stay on their
side stream of normal priority) against the default.  python scratch/prio_train.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")


def run(tag, main_priority):
    ctx = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=main_priority)) if main_priority is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        tr = bench.make_trainer("stage1", 1, dev, 0)
        pool = bench.train_pool(dev, 0, 1, 4)
        for rep in range(2):
            dt, host_ms, ls, _, _ = bench.timed_train_steps(tr, pool, 20, 6 if rep == 0 else 2, 1, dev)
            print("%-28s rep %d: %.3f ms/step (host %.2f)" % (tag, rep, dt / 20 * 1e3, host_ms), flush=True)
    finally:
        if ctx is not None:
            torch.cuda.synchronize()
            ctx.__exit__(None, None, None)


print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
run("default stream", None)
run("main stream priority -1", -1)
run("main stream priority 0 (new)", 0)
