"""CPU restatement (numpy, integer/byte arithmetic) of the image operations the reference's episode
pipeline performs through torchvision 0.7 / Pillow -- TEST INFRASTRUCTURE ONLY (imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline; never by the product path).

Reference call sites (data_kits/pascal_voc.py): :144 ``F.resize(x, size, Image.BILINEAR)`` on RGB images,
:145 ``F.resize(x, size, Image.NEAREST)`` on label images, :141-142 ``ToTensor`` + ``Normalize(mean, std)``,
:143 ``F.hflip``, :146 ``ColorJitter(0.4, 0.4, 0.4)``, :209-210/:231 ``mask // 255``.  The arithmetic lives in
third-party code that is NOT in /root/reference: torchvision (pinned 0.7.0, README.md:26) forwards to
Pillow (unpinned; any 7.x..12.x -- the resampling code has not changed): ``Image.resize`` ->
``ImagingResample`` (src/libImaging/Resample.c) for BILINEAR, ``ImagingTransform``/``ImagingScaleAffine``
(Geometry.c) for NEAREST, ``ImageEnhance`` -> ``Image.blend`` (Blend.c) and ``convert("L")`` (Convert.c)
for the colour jitter.  Their published algorithms are restated below; parity is pinned by fixtures
produced with Pillow itself (tests/golden/make_pil_golden.py -> tests/golden/pil_ops.npz).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: fixed-point fraction bits of the 8-bit path


def _bilinear_filter(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc (Resample.c) for the full box (0, in_size), BILINEAR.
    -> (ksize, bounds int32 [out,2] = (xmin, count), coeffs int32 [out,ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis0(img, out_size):
    """One 8-bit resampling pass along axis 0 of a [n, m, c] uint8 array."""
    ksize, bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for yy in range(out_size):
        y0, n = bounds[yy]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(n):
            acc += src[y0 + k] * int(kk[yy, k])
        out[yy] = _clip8(acc)
    return out


def resize_bilinear(img, out_h, out_w):
    """Image.resize((out_w, out_h), Image.BILINEAR) on an HWC uint8 image: horizontal pass, then vertical
    pass, each rounding to uint8 (ImagingResample)."""
    x = img
    if out_w != x.shape[1]:
        x = _resample_axis0(x.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if out_h != x.shape[0]:
        x = _resample_axis0(x, out_h)
    return np.ascontiguousarray(x)


def nearest_index(in_size, out_size):
    """ImagingScaleAffine (Geometry.c): source index of every output index for a pure scale.  The source
    coordinate is ACCUMULATED in double precision (xo = scale/2; xo += scale per output pixel), which is what
    decides the exact-boundary cases (e.g. output 200 of 500 -> 401)."""
    scale = in_size / out_size
    xo = scale * 0.5
    idx = np.empty(out_size, np.int64)
    for i in range(out_size):
        idx[i] = min(int(xo), in_size - 1)
        xo += scale
    return idx


def resize_nearest(img, out_h, out_w):
    """Image.resize((out_w, out_h), Image.NEAREST) on an HW (or HWC) uint8 image."""
    return img[nearest_index(img.shape[0], out_h)][:, nearest_index(img.shape[1], out_w)]


def to_tensor_normalize(img, mean, std):
    """ToTensor + Normalize: HWC uint8 -> CHW float32, ((x / 255) - mean) / std in float32 arithmetic."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


# ---- ColorJitter pieces (ImageEnhance.{Brightness,Contrast,Color} -> Image.blend) ---------------------------
def to_gray(img):
    """convert("L") (Convert.c rgb2l): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    x = img.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, img, alpha):
    """Image.blend(im1=degenerate, im2=img, alpha) (Blend.c): float32 interpolation, truncated (alpha in [0,1])
    or clipped extrapolation otherwise."""
    a = np.float32(alpha)
    t = degenerate.astype(np.float32) + a * (img.astype(np.float32) - degenerate.astype(np.float32))
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t)).astype(np.uint8)


def adjust_brightness(img, factor):
    return blend(np.zeros_like(img), img, factor)


def contrast_mean(img):
    """int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)"""
    g = to_gray(img)
    return int(float(g.astype(np.int64).sum()) / g.size + 0.5)


def adjust_contrast(img, factor):
    return blend(np.full_like(img, contrast_mean(img)), img, factor)


def adjust_saturation(img, factor):
    g = to_gray(img)
    return blend(np.repeat(g[..., None], 3, axis=2), img, factor)


def color_jitter(img, order, factors):
    """ColorJitter with a given draw: ``order`` a permutation of (0,1,2) = (brightness, contrast, saturation),
    ``factors`` the three enhancement factors."""
    fns = (adjust_brightness, adjust_contrast, adjust_saturation)
    for t in order:
        img = fns[t](img, factors[t])
    return img


def hflip(img):
    return np.ascontiguousarray(img[:, ::-1])


def support_mask_planes(mask_u8):
    """pascal_voc.py:209-210: (mask // 255) as float32 fg plane, 1 - fg as bg plane -> [2,H,W]."""
    fg = (mask_u8 // 255).astype(np.float32)
    return np.stack((fg, 1 - fg), axis=0)
