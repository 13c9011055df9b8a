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
        """Host path, same contract as the reference: pred/ref uint8-castable [B,H,W], cls iterable."""
        pred = np.asarray(pred, np.uint8)
        ref = np.asarray(ref, np.uint8)
        for i, ci in enumerate(cls):
            p, r = pred[i], ref[i]
            valid = r != 255
            for j, c in enumerate((0, int(ci))):
                tp = int(((p == j) & (r == j) & valid).sum())
                fp = int(((p == j) & (r != j) & valid).sum())
                fn = int(((p != j) & (r == j) & valid).sum())
                if verbose:
                    print(tp / (tp + fp + fn))
                self.stat[c] += (tp, fp, fn)

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


class Accumulator:
    """Running means of scalars / lists (reference: core/metrics.py:38-72)."""

    def __init__(self, **kwargs):
        for v in kwargs.values():
            if not isinstance(v, (float, int, list)):
                raise TypeError(f"The Accumulator does not support `{type(v)}`. Supported types: [float, int, list]")
        self.values = kwargs
        self.counter = {k: 0 for k in kwargs}

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(self.values[k], list):
                self.values[k].append(v)
            else:
                self.values[k] = self.values[k] + v
            self.counter[k] += 1

    def mean(self, key, axis=None):
        if not isinstance(key, str):
            return [self.mean(k, axis) for k in key]
        v = self.values[key]
        return np.array(v).mean(axis) if isinstance(v, list) else v / self.counter[key]

    def std(self, key, axis=None):
        if isinstance(key, str):
            if isinstance(self.values[key], list):
                return np.array(self.values[key]).std(axis)
            raise RuntimeError("`std` is not supported for (int, float). Use list instead.")
        if isinstance(key, (list, tuple)):
            return [self.std(k, axis) for k in key]
        raise TypeError(f"`key` must be a str/list/tuple, got {type(key)}")
