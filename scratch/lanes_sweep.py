"""The reference's 5 x 1000 single-episode protocol (bench.protocol_5x1000) against the number of steps in flight.
python scratch/lanes_sweep.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
args = types.SimpleNamespace(dataset="PASCAL")
run = bench.EvalRunner(dev, 0, "stage1", 1, 25, "PASCAL", 2, graph=True, loss="ce")
for lanes in (2, 4, 6, 8, 12):
    for exact in (False,):
        r = bench.protocol_5x1000(run.net, run.pool, dev, args, rounds=2, test_n=1000, lanes=lanes)
        print("lanes %2d: %.1f episodes/s (wall %.2f s), mIoU %s" % (lanes, r["episodes_per_s"], r["wall_s"], r["miou_per_round"]), flush=True)
