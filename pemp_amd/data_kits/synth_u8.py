"""Deterministic synthetic "decoded JPEG / PNG label" arrays (uint8) for the episode input pipeline:
no dataset exists on either box, so tests and bench.py feed the device-side preprocessing with these.
Pure integer hashing (splitmix64, as pemp_amd/synth.py) -- identical on every machine."""
import numpy as np

_M = (1 << 64) - 1


def _mix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M)
    return z ^ (z >> np.uint64(31))


def image(seed, h, w):
    """HWC uint8 RGB: blocky low-frequency texture + per-pixel noise (exercises every filter tap)."""
    with np.errstate(over="ignore"):
        yy, xx, cc = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64),
                                 np.arange(3, dtype=np.uint64), indexing="ij")
        coarse = _mix((yy // np.uint64(7)) * np.uint64(1315423911) + (xx // np.uint64(9)) * np.uint64(2654435761)
                      + cc * np.uint64(97) + np.uint64(seed) * np.uint64(1000003))
        fine = _mix(yy * np.uint64(40503) + xx * np.uint64(69069) + cc * np.uint64(7) + np.uint64(seed + 1) * np.uint64(7919))
    v = (coarse % np.uint64(200)).astype(np.int64) + (fine % np.uint64(56)).astype(np.int64)
    return v.astype(np.uint8)


def mask(seed, h, w):
    """HW uint8 label image with values {0, 255}: two ellipses + a sprinkle of isolated pixels."""
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    with np.errstate(over="ignore"):
        r = int(_mix(np.array([seed], np.uint64))[0] % np.uint64(1000))
    cy, cx = h * (0.3 + 0.4 * (r % 10) / 10), w * (0.3 + 0.4 * (r // 10 % 10) / 10)
    m = ((yy - cy) / (0.22 * h)) ** 2 + ((xx - cx) / (0.18 * w)) ** 2 < 1.0
    m |= ((yy - 0.75 * h) / (0.1 * h)) ** 2 + ((xx - 0.2 * w) / (0.12 * w)) ** 2 < 1.0
    with np.errstate(over="ignore"):
        spr = _mix(np.arange(h * w, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)).reshape(h, w)
    m ^= (spr % np.uint64(37)) == 0
    return (m.astype(np.uint8) * 255)
