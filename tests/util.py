"""Helpers shared by the parity tests."""
import json
import os

import numpy as np
import torch

from pemp_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Shape:
    def __init__(self, shape):
        self.shape = tuple(shape)


def key_spec(name):
    with open(os.path.join(GOLD, f"state_keys_{name}.json")) as f:
        return json.load(f)


def wgen_state_dict(name, seed=1234):
    """Wgen weights for the reference key layout `name` as {key: torch tensor}."""
    spec = key_spec(name)
    sd = synth.gen_state_dict({k: _Shape(s) for k, s, _ in spec}, seed)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def episode_tensors(seed, shot, H, out_hw, device=None):
    ep = synth.make_episode(int(seed), shot=int(shot), height=int(H), width=int(H), out_hw=tuple(int(v) for v in out_hw))
    t = {k: torch.from_numpy(v)[None] for k, v in ep.items() if k != "cls"}
    t["qry_mask"] = t["qry_mask"][0]          # [1,Ho,Wo]
    if device is not None:
        t = {k: v.to(device) for k, v in t.items()}
    t["cls"] = ep["cls"]
    return t


def counts(pred, ref):
    out = []
    v = ref != 255
    for j in (0, 1):
        out.append([int(((pred == j) & (ref == j) & v).sum()), int(((pred == j) & (ref != j) & v).sum()),
                    int(((pred != j) & (ref == j) & v).sum())])
    return np.array(out, np.int64)
