"""Seeded INPUT cases shared by the fixture generator (make_golden.py, which alone imports the reference) and the tests:
inputs only -- no reference code, nothing that needs /root/reference."""
import numpy as np
import torch

from pemp_amd import synth


def stage2_train_prior(qry_mask):
    """Deterministic stand-in for the stage-1 argmax prior of the stage-2 train-step fixture: the query
    foreground shifted by (3, 5) pixels (the real prior path is pinned by the stage-2 eval fixtures)."""
    fg = (qry_mask[:, 0] == 1)
    return np.roll(fg, (3, 5), axis=(1, 2)).astype(np.int64)[:, None]          # [BQ,1,H,W]


def cedt_cases():
    """Targets / logits of the CELossDT fixture (shared with the tests): two 97x97 query masks (one with an ignored
    corner), one 333x500 mask, an all-background and an all-foreground map; logits from a seeded generator."""
    ts = [torch.from_numpy(synth.make_episode(s, out_hw=hw)["qry_mask"][0]) for s, hw in ((41, (97, 97)), (42, (97, 97)))]
    t = torch.stack(ts)
    t[0, :4, :9] = 255
    cases = [t, torch.from_numpy(synth.make_episode(43, out_hw=(333, 500))["qry_mask"]),
             torch.zeros(1, 40, 57, dtype=torch.int64), torch.ones(1, 9, 11, dtype=torch.int64)]
    g = torch.Generator().manual_seed(77)
    return [(c, torch.rand(c.shape[0], 2, *c.shape[-2:], generator=g) * 6 - 3) for c in cases]


def metric_cases():
    rng = np.random.RandomState(0)
    out = []
    for cls in (1, 3, 3, 5, 17):
        pred = rng.randint(0, 2, (1, 40, 50))
        ref = rng.randint(0, 2, (1, 40, 50))
        ref[0, :3] = 255
        out.append((pred, ref, [cls]))
    return out


def metric_cases_coco():
    """COCO-20i, split 1: labels 21..40 of an [81, 3] table; every label once, two of them twice, one ignored band."""
    rng = np.random.RandomState(1)
    out = []
    for cls in list(range(21, 41)) + [21, 40]:
        pred = rng.randint(0, 2, (1, 48, 64))
        ref = rng.randint(0, 2, (1, 48, 64))
        ref[0, -2:] = 255
        out.append((pred, ref, [cls]))
    return out
