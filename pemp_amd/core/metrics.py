"""Few-shot segmentation metrics (reference: core/metrics.py:4-35) with a device-side feed.

The tp/fp/fn table is integer-valued, so tables accumulated on different ranks can be summed
exactly (one all-reduce per evaluation round) and the resulting mIoU is
bit-identical to a single-process run.
"""
import numpy as np


class FewShotMetric:
    def __init__(self, classes):
        self.classes = classes
        self.stat = np.zeros((classes + 1, 3))          # rows: bg, class 1..C; cols: tp, fp, fn

    def update(self, pred, ref, cls, verbose=0):
        """Host path, same contract as the reference (core/metrics.py:9-23): pred / ref uint8-castable [B,H,W], cls iterable;
        pixels labelled 255 are ignored; per episode the background row and the episode's class row receive (tp, fp, fn)."""
        pred = np.asarray(pred, np.uint8)
        ref = np.asarray(ref, np.uint8)
        for p, r, ci in zip(pred, ref, cls):
            keep = r != 255
            p, r = p[keep], r[keep]
            for value, row in ((0, 0), (1, int(ci))):
                hit_p, hit_r = p == value, r == value
                counts = (int((hit_p & hit_r).sum()), int((hit_p & ~hit_r).sum()), int((~hit_p & hit_r).sum()))
                if verbose:
                    print(counts[0] / sum(counts))
                self.stat[row] += counts

    def update_counts(self, counts, cls):
        """Device path: counts [B,6] = (tp,fp,fn) for bg then fg, as produced by pemp_eval_tail_f32."""
        counts = np.asarray(counts, np.float64).reshape(-1, 2, 3)
        for i, ci in enumerate(cls):
            self.stat[0] += counts[i, 0]
            self.stat[int(ci)] += counts[i, 1]

    def mIoU(self, labels, binary=False):
        if binary:
            stat = np.c_[self.stat[0], self.stat[1:].sum(axis=0)].T
        else:
            stat = self.stat[labels]
        tp, fp, fn = stat.T
        per_class = tp / (tp + fp + fn)
        return per_class, per_class.mean()


class _Series:
    """Values appended one by one (``loss=[]``): mean / std over the collected array along ``axis``."""

    def __init__(self, start):
        self.items = list(start)

    def add(self, v):
        self.items.append(v)

    def mean(self, axis, _count):
        return np.array(self.items).mean(axis)

    def std(self, axis):
        return np.array(self.items).std(axis)


class _Sum:
    """A running sum (``n=0.0``): mean = sum / number of updates; no standard deviation."""

    def __init__(self, start):
        self.items = start

    def add(self, v):
        self.items = self.items + v

    def mean(self, _axis, count):
        return self.items / count

    def std(self, _axis):
        raise RuntimeError("`std` is not supported for (int, float). Use list instead.")


class Accumulator:
    """Per-key running statistics with the contract of the reference's class (core/metrics.py:38-72): a key declared with a list
    collects what it is fed, a key declared with a number sums it; ``mean(key | keys, axis)``, ``std(...)``, ``.values`` /
    ``.counter`` as there."""

    def __init__(self, **declared):
        self._slots = {}
        for key, start in declared.items():
            if isinstance(start, list):
                self._slots[key] = _Series(start)
            elif isinstance(start, (float, int)):
                self._slots[key] = _Sum(start)
            else:
                raise TypeError(f"The Accumulator does not support `{type(start)}`. Supported types: [float, int, list]")
        self.counter = dict.fromkeys(declared, 0)

    @property
    def values(self):
        return {key: slot.items for key, slot in self._slots.items()}

    def update(self, **fed):
        for key, v in fed.items():
            self._slots[key].add(v)
            self.counter[key] += 1

    def _each(self, key, fn):
        if isinstance(key, str):
            return fn(key)
        if isinstance(key, (list, tuple)):
            return [fn(k) for k in key]
        raise TypeError(f"`key` must be a str/list/tuple, got {type(key)}")

    def mean(self, key, axis=None):
        return self._each(key, lambda k: self._slots[k].mean(axis, self.counter[k]))

    def std(self, key, axis=None):
        return self._each(key, lambda k: self._slots[k].std(axis))
