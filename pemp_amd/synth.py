"""Deterministic synthetic weights and episodes (``Wgen`` and ``E(seed)``).

Neither box has the PASCAL/COCO datasets or any checkpoint, so parity and throughput are
measured on synthetic episodes with synthetic weights.  Everything here is a pure function of
integer seeds, built from IEEE-exact operations only (integer hashing, +, -, *, floor, abs,
comparisons) -- no libm transcendental, no torch/numpy RNG stream -- so that the container that
wrote ``tests/golden`` and the GPU box that checks it derive bit-identical tensors.

Shapes and dtypes follow the reference's data layer:
  sup_img  [B,S,3,H,W] f32, sup_mask [B,S,2,H,W] f32 (ch0 = fg, ch1 = 1-fg;
  data_kits/pascal_voc.py:209-210), qry_img [B,Q,3,H,W] f32, qry_mask [B,Q,Ho,Wo] int64 at the
  query's ORIGINAL size (pascal_voc.py:229), cls int64 (split*5+1 .. split*5+5, datasets.py:98).
"""
import zlib

import numpy as np

_U = np.uint64
_GOLD = _U(0x9E3779B97F4A7C15)
_M1 = _U(0xBF58476D1CE4E5B9)
_M2 = _U(0x94D049BB133111EB)

#: query ground-truth sizes cycled by episode index
QUERY_SIZES = ((333, 500), (375, 500), (457, 500), (366, 500), (500, 333))
#: COCO-20i: the val2014 picture formats, up to 640x640; test-time query labels keep the
#: picture's own size (data_kits/coco.py, as pascal_voc.py:229)
QUERY_SIZES_COCO = ((480, 640), (640, 480), (427, 640), (640, 640), (375, 500), (640, 427), (512, 640), (500, 333))


def query_sizes(dataset="PASCAL"):
    if dataset == "PASCAL":
        return QUERY_SIZES
    if dataset == "COCO":
        return QUERY_SIZES_COCO
    raise ValueError(f"Not supported dataset: {dataset}. [PASCAL, COCO]")


def val_labels(split, dataset="PASCAL"):
    """Validation classes of a split (data_kits/datasets.py:83-104): 5 per PASCAL-5i split, 20 per COCO-20i split."""
    if dataset == "PASCAL":
        return list(range(split * 5 + 1, split * 5 + 6))
    if dataset == "COCO":
        return list(range(split * 20 + 1, split * 20 + 21))
    raise ValueError(f"Not supported dataset: {dataset}. [PASCAL, COCO]")


def num_classes(dataset="PASCAL"):
    """Rows of the metric table minus background (entry/pemp_stage1.py:151: 20 for PASCAL, 80 for COCO)."""
    return {"PASCAL": 20, "COCO": 80}[dataset]


def _mix(z):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    z = (z + _GOLD).astype(np.uint64)
    z = ((z ^ (z >> _U(30))) * _M1).astype(np.uint64)
    z = ((z ^ (z >> _U(27))) * _M2).astype(np.uint64)
    return z ^ (z >> _U(31))


def _key(seed, name):
    k = np.array([(int(seed) & 0xFFFFFFFF) << 32 | zlib.crc32(name.encode())], dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix(_mix(k))[0]


def uniform01(seed, name, n):
    """n float64 values in [0,1) with 24 random bits each; function of (seed, name, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        bits = _mix((idx * _GOLD).astype(np.uint64) ^ _key(seed, name))
    return (bits >> _U(40)).astype(np.float64) * (1.0 / 16777216.0)


def uniform(seed, name, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------------
# Wgen: weights
# --------------------------------------------------------------------------------------------
def _isqrt_scale(fan_in):
    # sqrt via Newton on float64 is correctly rounded by IEEE sqrt; np.sqrt is exact (IEEE op).
    return float(np.sqrt(6.0 / float(fan_in)))


def gen_tensor(seed, name, shape, kind):
    """One parameter/buffer.  ``kind`` selects the distribution."""
    if kind == "conv_w":          # He-uniform over fan_in = Cin*KH*KW
        a = _isqrt_scale(int(np.prod(shape[1:])))
        return uniform(seed, name, shape, -a, a)
    if kind == "linear_w":
        a = _isqrt_scale(shape[1]) * 0.5
        return uniform(seed, name, shape, -a, a)
    if kind == "bias":
        return uniform(seed, name, shape, -0.1, 0.1)
    if kind == "bn_gamma":
        return uniform(seed, name, shape, 0.5, 1.5)
    if kind == "bn_gamma_res":    # last BN of a residual branch: damped so 13 blocks stay tame
        return uniform(seed, name, shape, 0.2, 0.6)
    if kind == "bn_beta":
        return uniform(seed, name, shape, -0.1, 0.1)
    if kind == "bn_mean":
        return uniform(seed, name, shape, -0.1, 0.1)
    if kind == "bn_var":
        return uniform(seed, name, shape, 0.5, 1.5)
    if kind == "ctr":             # torch.rand-like, networks/pemp_stage1.py:105
        return uniform(seed, name, shape, 0.0, 1.0)
    raise ValueError(kind)


def classify(name, shape):
    """Map a state_dict key + shape to a generator kind (keys as in SURVEY.md §8 a13)."""
    leaf = name.rsplit(".", 1)[-1]
    if name == "ctr":
        return "ctr"
    if leaf == "num_batches_tracked":
        return None
    if leaf == "running_mean":
        return "bn_mean"
    if leaf == "running_var":
        return "bn_var"
    if len(shape) == 4:
        return "conv_w"
    if len(shape) == 2:
        return "linear_w"
    if leaf == "bias":
        # BN beta vs conv bias: BN modules also own running_mean, decided by the caller
        return "bias"
    if leaf == "weight" and len(shape) == 1:
        return "bn_gamma_res" if (".bn3." in name) else "bn_gamma"
    raise ValueError(f"cannot classify {name} {shape}")


def gen_state_dict(template, seed=1234):
    """Fill every entry of ``template`` (name -> array-like with .shape) deterministically.

    Returns {name: np.ndarray}; ``num_batches_tracked`` entries are int64 zeros.
    """
    names = list(template.keys())
    bn_prefixes = {n.rsplit(".", 1)[0] for n in names if n.endswith("running_mean")}
    out = {}
    for n in names:
        shape = tuple(template[n].shape)
        kind = classify(n, shape)
        if kind is None:
            out[n] = np.zeros(shape, dtype=np.int64)
            continue
        if kind == "bias" and n.rsplit(".", 1)[0] in bn_prefixes:
            kind = "bn_beta"
        out[n] = gen_tensor(seed, n, shape, kind)
    return out


# --------------------------------------------------------------------------------------------
# E(seed): episodes
# --------------------------------------------------------------------------------------------
def _tri(t):
    """Triangle wave with period 1 and range [-1,1]; exact arithmetic only."""
    return 4.0 * np.abs(t - np.floor(t + 0.5)) - 1.0


def _texture(seed, name, yy, xx, fmin, fmax, offset):
    """[3,H,W] float64 texture: 4 triangle-wave plane waves per channel + per-channel offset."""
    p = uniform01(seed, name, 3 * 4 * 4).reshape(3, 4, 4)
    out = np.empty((3,) + yy.shape, dtype=np.float64)
    for c in range(3):
        acc = np.zeros_like(yy)
        for k in range(4):
            fx = fmin + (fmax - fmin) * p[c, k, 0]
            fy = fmin + (fmax - fmin) * p[c, k, 1]
            amp = 0.25 + 0.5 * p[c, k, 3]
            acc = acc + amp * _tri(fx * xx + fy * yy + p[c, k, 2])
        out[c] = acc * 0.6 + offset[c]
    return out


def _ellipses(seed, name):
    p = uniform01(seed, name, 1 + 3 * 4)
    n = 1 + int(p[0] * 3.0)
    e = p[1:].reshape(3, 4)[:n]
    cy = 0.25 + 0.5 * e[:, 0]
    cx = 0.25 + 0.5 * e[:, 1]
    ry = 0.10 + 0.18 * e[:, 2]
    rx = 0.10 + 0.18 * e[:, 3]
    return cy, cx, ry, rx


def _raster(ell, h, w):
    cy, cx, ry, rx = ell
    yy = ((np.arange(h, dtype=np.float64) + 0.5) / h)[:, None]
    xx = ((np.arange(w, dtype=np.float64) + 0.5) / w)[None, :]
    m = np.zeros((h, w), dtype=bool)
    for i in range(len(cy)):
        dy = (yy - cy[i]) / ry[i]
        dx = (xx - cx[i]) / rx[i]
        m |= (dy * dy + dx * dx) <= 1.0
    return m


def _image(seed, tag, mask, h, w):
    yy = ((np.arange(h, dtype=np.float64) + 0.5) / h)[:, None] * np.ones((1, w))
    xx = ((np.arange(w, dtype=np.float64) + 0.5) / w)[None, :] * np.ones((h, 1))
    off = uniform01(seed, "coloff", 6) - 0.5
    bg = _texture(seed, "bgtex", yy, xx, 1.0, 5.0, off[:3] * 1.2)
    fg = _texture(seed, "fgtex", yy, xx, 7.0, 17.0, off[3:] * 1.2 + 0.6)
    noise = (uniform01(seed, "noise" + tag, 3 * h * w).reshape(3, h, w) - 0.5) * 0.3
    img = np.where(mask[None], fg, bg) + noise
    return img.astype(np.float32)


def make_episode(seed, shot=1, height=401, width=401, index=None, out_hw=None, split=0, dataset="PASCAL"):
    """One 1-way ``shot``-shot episode.

    Returns dict of numpy arrays WITHOUT the batch dim:
      sup_img [S,3,H,W] f32, sup_mask [S,2,H,W] f32, qry_img [1,3,H,W] f32,
      qry_mask [1,Ho,Wo] int64 in {0,1}, cls int.
    Query GT size cycles over the dataset's size table by ``index`` (default: seed) unless ``out_hw`` given; the
    class cycles over the split's validation labels by ``seed`` (every label of a split receives episodes).
    """
    if index is None:
        index = seed
    sizes = query_sizes(dataset)
    ho, wo = out_hw if out_hw is not None else sizes[int(index) % len(sizes)]
    sup_img = np.empty((shot, 3, height, width), np.float32)
    sup_mask = np.empty((shot, 2, height, width), np.float32)
    for s in range(shot):
        ell = _ellipses(seed, f"sup{s}")
        m = _raster(ell, height, width)
        sup_img[s] = _image(seed, f"s{s}", m, height, width)
        sup_mask[s, 0] = m
        sup_mask[s, 1] = ~m
    qell = _ellipses(seed, "qry")
    qm = _raster(qell, height, width)
    qry_img = _image(seed, "q", qm, height, width)[None]
    qry_mask = _raster(qell, ho, wo).astype(np.int64)[None]
    labels = val_labels(max(int(split), 0), dataset)
    cls = labels[int(seed) % len(labels)]
    return dict(sup_img=sup_img, sup_mask=sup_mask, qry_img=qry_img, qry_mask=qry_mask, cls=cls)


def make_batch(seeds, shot=1, height=401, width=401, out_hw=None, split=0, dataset="PASCAL"):
    """Stack episodes along B.  All query masks must share one size (pass ``out_hw``)."""
    eps = [make_episode(s, shot, height, width, out_hw=out_hw, split=split, dataset=dataset) for s in seeds]
    return {k: (np.stack([e[k] for e in eps]) if k != "cls" else np.array([e[k] for e in eps]))
            for k in eps[0]}


# --------------------------------------------------------------------------------------------
# Wgen for a module: the state_dict itself is the template (keys / shapes are the reference's, pinned by
# tests/golden/state_keys_*.json), every tensor a function of (seed, key, shape) only
# --------------------------------------------------------------------------------------------
def wgen_state_dict_for(module, seed=1234):
    """Wgen weights for ``module`` (any torch module with the reference's key layout) as {key: torch tensor}."""
    import torch
    sd = gen_state_dict(module.state_dict(), seed)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
