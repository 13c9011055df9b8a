"""Device-side episode preprocessing: the part of ``PascalVOCTrain._get_episode`` (reference
data_kits/pascal_voc.py:185-237) that follows the JPEG/PNG decode, moved to the GPU.

Host work left: decode (PIL), the random draws of the augmentation (Python ``random`` in the reference's
order, so a seeded run consumes the generator exactly as the reference does) and ``crop_obj``'s window
choice (:26-84, needs a handful of counts on the label image).  Everything that touches pixels --
bilinear / nearest resize, colour jitter, flip, crop, ToTensor, Normalize, ``mask // 255`` planes -- runs in
``pemp_episode_preprocess`` on uint8 data uploaded as ONE pinned blob per batch (descriptors + pixels):
a 375x500 JPEG travels as 0.56 MB instead of the 1.93 MB fp32 tensor the reference's loader ships.

``EpisodeLoader`` double-buffers: upload + preprocessing of batch k+1 run on a side HIP stream while
the model consumes batch k.
"""
import ctypes as C
import random as _random

import numpy as np
import torch

from .. import _lib
from .._lib import SampleDesc

MEAN = (0.485, 0.456, 0.406)        # data_kits/datasets.py:18-19
STD = (0.229, 0.224, 0.225)
_OPS = {"brightness": 1, "contrast": 2, "saturation": 3}


def _align(v, a=16):
    return (v + a - 1) // a * a


class Sample:
    """One decoded image (+ label) and what to do with it."""
    __slots__ = ("img", "mask", "mask_mode", "scaled", "crop", "flip", "jitter")

    def __init__(self, img=None, mask=None, mask_mode=0, scaled=None, crop=(0, 0), flip=False, jitter=None):
        self.img, self.mask, self.mask_mode = img, mask, mask_mode
        self.scaled, self.crop, self.flip, self.jitter = scaled, crop, flip, jitter


# -- host mirror of the reference's random draws ------------------------------------------------------------
def nearest_index(in_size, out_size):
    """Source index table of Pillow's NEAREST resize (coordinate accumulated in double, Geometry.c)."""
    scale = in_size / out_size
    xo = scale * 0.5
    idx = np.empty(out_size, np.int64)
    for i in range(out_size):
        idx[i] = min(int(xo), in_size - 1)
        xo += scale
    return idx


def draw_jitter(rng, strength=0.4):
    """ColorJitter(brightness=s, contrast=s, saturation=s).get_params of torchvision 0.7: three uniform
    factors, then ``random.shuffle`` of the op list.  -> (order tuple of op names, factors by name)."""
    lo, hi = max(0.0, 1.0 - strength), 1.0 + strength
    factors = {k: rng.uniform(lo, hi) for k in ("brightness", "contrast", "saturation")}
    order = ["brightness", "contrast", "saturation"]
    rng.shuffle(order)
    return tuple(order), factors


def crop_obj_origin(mask, height, width, rng):
    """Window origin chosen by ``crop_obj`` (data_kits/pascal_voc.py:26-84) on the resized + flipped uint8
    label image: a random window, re-drawn around the object (or around the background) when the window
    holds fewer than 1024 of its pixels; same retry rule and same draw order as the reference."""
    mh, mw = mask.shape

    def window(my, mx):
        return mask[my:my + height, mx:mx + width]

    def around(m):
        ys = np.where(m.max(axis=1) > 0)[0]
        xs = np.where(m.max(axis=0) > 0)[0]
        ymin, ymax, xmin, xmax = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
        y0 = max(0, ymax - height)
        y1 = max(min(mh - height, ymin), y0)
        x0 = max(0, xmax - width)
        x1 = max(min(mw - width, xmin), x0)
        return rng.randint(y0, y1), rng.randint(x0, x1)

    def retry(my, mx):
        for _ in range(102):
            my, mx = rng.randint(0, mh - height), rng.randint(0, mw - width)
            if np.count_nonzero(window(my, mx)) > 0:
                break
        return my, mx

    my, mx = rng.randint(0, mh - height), rng.randint(0, mw - width)
    if np.count_nonzero(window(my, mx)) < 1024:                       # small foreground
        my, mx = around(mask)
        if np.count_nonzero(window(my, mx)) == 0:
            my, mx = retry(my, mx)
    elif np.count_nonzero(255 - window(my, mx)) < 1024:               # small background
        my, mx = around(255 - mask)
        if np.count_nonzero(255 - window(my, mx)) == 0:
            my, mx = retry(my, mx)
    return int(my), int(mx)


def train_samples(sup, qry, height, width, rng=_random):
    """Augmentation draws of one training episode in the reference's order (pascal_voc.py:194-226).
    sup / qry: lists of (img uint8 HWC, label uint8 HW).  -> list of Sample (supports first)."""
    out = []
    for img, lab in sup:
        f = rng.uniform(1, 1.5)
        sh, sw = int(height * f), int(width * f)
        flag = rng.random()
        order, fac = draw_jitter(rng)
        flip = flag >= 0.5
        m = lab[nearest_index(lab.shape[0], sh)][:, nearest_index(lab.shape[1], sw)]
        if flip:
            m = m[:, ::-1]
        oy, ox = crop_obj_origin(m, height, width, rng)
        out.append(Sample(img, lab, 1, (sh, sw), (oy, ox), flip, (order, fac)))
    for img, lab in qry:
        flag = rng.random()
        order, fac = draw_jitter(rng)
        out.append(Sample(img, lab, 2, (height, width), (0, 0), flag >= 0.5, (order, fac)))
    return out


def test_samples(sup, qry, height, width):
    """Evaluation episode (pascal_voc.py:202-206,221-226): resize only; the query label keeps its size."""
    return [Sample(img, lab, 1, (height, width)) for img, lab in sup] + \
           [Sample(img, lab, 3, (height, width)) for img, lab in qry]


# -- staging + launch ----------------------------------------------------------------------------------------
class StagedBatch:
    __slots__ = ("descs", "n", "nbytes", "ws_bytes", "n_img", "n_planes", "label_shapes", "label_elems", "host", "pixel_bytes")


class EpisodeTransform:
    def __init__(self, height=401, width=401, mean=MEAN, std=STD, device=None):
        self.H, self.W = height, width
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ws = None

    def stage(self, samples, host=None):
        """Fill a pinned host buffer with descriptors + pixels.  -> StagedBatch (host buffer inside)."""
        lib = _lib.load()
        H, W, n = self.H, self.W, len(samples)
        descs = (SampleDesc * n)()
        off = _align(C.sizeof(SampleDesc) * n, 256)
        n_img = n_pl = 0
        lab_off, lab_shapes = 0, []
        copies = []
        for d, s in zip(descs, samples):
            d.img_off = d.msk_off = -1
            if s.img is not None:
                if s.img.dtype != np.uint8 or s.img.ndim != 3 or s.img.shape[2] != 3:
                    raise ValueError("Sample.img must be HWC uint8 RGB")
                d.hs, d.ws = s.img.shape[:2]
                d.img_off, d.img_out = off, n_img * 3 * H * W
                copies.append((off, s.img))
                off = _align(off + s.img.size)
                n_img += 1
            if s.mask is not None:
                if s.mask.dtype != np.uint8 or s.mask.ndim != 2 or (s.img is not None and s.mask.shape != s.img.shape[:2]):
                    raise ValueError("Sample.mask must be HW uint8 with the image's size")
                d.hs, d.ws = s.mask.shape
                d.msk_off = off
                copies.append((off, s.mask))
                off = _align(off + s.mask.size)
            d.mask_mode = s.mask_mode
            d.sh, d.sw = s.scaled if s.scaled is not None else (H, W)
            d.oy, d.ox = s.crop
            d.flip = int(bool(s.flip))
            if s.jitter is not None:
                order, fac = s.jitter
                d.jitter_order = sum(_OPS[name] << (2 * i) for i, name in enumerate(order))
                d.jitter[0], d.jitter[1], d.jitter[2] = fac["brightness"], fac["contrast"], fac["saturation"]
            if s.mask_mode == 1:
                d.msk_out = n_pl * 2 * H * W
                n_pl += 1
            elif s.mask_mode in (2, 3):
                shape = (H, W) if s.mask_mode == 2 else tuple(s.mask.shape)
                d.msk_out = lab_off
                lab_shapes.append((lab_off, shape))
                lab_off += shape[0] * shape[1]
        ws_bytes = lib.pemp_episode_plan(descs, n, H, W)
        if ws_bytes == 0:
            _lib.check(-1, "episode_plan")
        if host is None or host.numel() < off:
            host = torch.empty(max(off, 1 << 20), dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        C.memmove(hv.ctypes.data, C.addressof(descs), C.sizeof(descs))
        for o, a in copies:
            hv[o:o + a.size] = a.reshape(-1)
        b = StagedBatch()
        b.descs, b.n, b.nbytes, b.ws_bytes, b.n_img, b.n_planes = descs, n, off, ws_bytes, n_img, n_pl
        b.label_shapes, b.label_elems, b.host = lab_shapes, lab_off, host
        b.pixel_bytes = off - _align(C.sizeof(SampleDesc) * n, 256)
        return b

    def run(self, b, blob=None):
        """Upload (async, pinned) and preprocess on the current stream.
        -> (images [n_img,3,H,W] fp32, planes [n_sup,2,H,W] fp32, labels list of int64 [h,w] views)."""
        lib = _lib.load()
        dev, H, W = self.device, self.H, self.W
        if blob is None or blob.numel() < b.nbytes:
            blob = torch.empty(b.host.numel(), dtype=torch.uint8, device=dev)
        blob[:b.nbytes].copy_(b.host[:b.nbytes], non_blocking=True)
        if self._ws is None or self._ws.numel() < b.ws_bytes:
            self._ws = torch.empty(int(b.ws_bytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        img = torch.empty((b.n_img, 3, H, W), dtype=torch.float32, device=dev)
        planes = torch.empty((b.n_planes, 2, H, W), dtype=torch.float32, device=dev)
        labels = torch.empty(max(b.label_elems, 1), dtype=torch.int64, device=dev)
        _lib.check(lib.pemp_episode_preprocess(blob.data_ptr(), b.descs, blob.data_ptr(), b.n, H, W, self.mean, self.std,
                                               img.data_ptr(), planes.data_ptr(), labels.data_ptr(), self._ws.data_ptr(),
                                               self._ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "episode_preprocess")
        self._blob = blob
        return img, planes, [labels[o:o + h * w].view(h, w) for o, (h, w) in b.label_shapes]

    def __call__(self, samples):
        return self.run(self.stage(samples))


class EpisodeLoader:
    """Iterates ``batches`` (an iterable of Sample lists), keeping one batch in flight on a side stream:
    while the caller computes on batch k, batch k+1 is staged on the host, uploaded and preprocessed."""

    def __init__(self, batches, transform, depth=2):
        self.it, self.tf = iter(batches), transform
        from .. import ops
        self.stream = ops.concurrent_stream(torch.device(transform.device))      # a stream that really runs beside the consumer's
        self.slots = [dict(host=None, blob=None, ws=None, tf=EpisodeTransform(transform.H, transform.W, device=transform.device))
                      for _ in range(depth + 1)]
        for s in self.slots:
            s["tf"].mean, s["tf"].std = transform.mean, transform.std
        self.k = 0
        self.queue = []
        for _ in range(depth):
            self._launch()

    def _launch(self):
        try:
            samples = next(self.it)
        except StopIteration:
            return
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        if "done" in slot:
            slot["done"].synchronize()            # the consumer of this slot's previous batch has been enqueued long ago
        staged = slot["tf"].stage(samples, slot["host"])
        slot["host"] = staged.host
        with torch.cuda.stream(self.stream):
            out = slot["tf"].run(staged, slot["blob"])
            slot["blob"] = slot["tf"]._blob
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["done"] = ev
        self.queue.append((out, ev))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        out, ev = self.queue.pop(0)
        torch.cuda.current_stream().wait_event(ev)
        for t in (out[0], out[1], *out[2]):
            t.record_stream(torch.cuda.current_stream())
        self._launch()
        return out
