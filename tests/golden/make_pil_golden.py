"""Fixtures for the episode input pipeline, produced by Pillow itself (the third-party library the
reference's data_kits/pascal_voc.py:141-146 calls through torchvision): tests/golden/pil_ops.npz.

    python tests/golden/make_pil_golden.py

Inputs are stored for the small cases; the full-size cases store the generator seed, a CRC32 of Pillow's
output and a strided sample, so the file stays small.
"""
import sys
import zlib
from pathlib import Path

import numpy as np
from PIL import Image, ImageEnhance

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pemp_amd.data_kits import synth_u8          # noqa: E402

SMALL = [(60, 83, 97, 97), (150, 200, 97, 97), (97, 97, 97, 97), (33, 140, 120, 61), (211, 97, 97, 130)]
FULL = [(375, 500, 401, 401), (500, 333, 401, 401), (640, 427, 401, 401), (366, 500, 601, 457)]
JITTER = [((0, 1, 2), (0.6, 1.4, 0.83)), ((2, 0, 1), (1.27, 0.71, 1.4)), ((1, 2, 0), (1.0, 1.0, 1.0)),
          ((2, 1, 0), (1.399, 0.601, 1.113))]


def crc(a):
    return np.array(zlib.crc32(np.ascontiguousarray(a).tobytes()), np.uint32)


def main():
    out = {}
    for i, (hs, ws, h, w) in enumerate(SMALL):
        img, msk = synth_u8.image(100 + i, hs, ws), synth_u8.mask(100 + i, hs, ws)
        out[f"s{i}_img"], out[f"s{i}_msk"], out[f"s{i}_hw"] = img, msk, np.array([h, w])
        out[f"s{i}_bilinear"] = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        out[f"s{i}_nearest"] = np.asarray(Image.fromarray(msk).resize((w, h), Image.NEAREST))
    for i, (hs, ws, h, w) in enumerate(FULL):
        img, msk = synth_u8.image(200 + i, hs, ws), synth_u8.mask(200 + i, hs, ws)
        b = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        n = np.asarray(Image.fromarray(msk).resize((w, h), Image.NEAREST))
        out[f"f{i}_dims"] = np.array([hs, ws, h, w])
        out[f"f{i}_bilinear_crc"], out[f"f{i}_nearest_crc"] = crc(b), crc(n)
        out[f"f{i}_bilinear_s"], out[f"f{i}_nearest_s"] = b[::13, ::11], n[::13, ::11]
    img = synth_u8.image(300, 90, 120)
    out["j_img"] = img
    im = Image.fromarray(img)
    enh = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)
    for i, (order, factors) in enumerate(JITTER):
        x = im
        for t in order:
            x = enh[t](x).enhance(factors[t])
        out[f"j{i}_order"], out[f"j{i}_factors"], out[f"j{i}_out"] = np.array(order), np.array(factors), np.asarray(x)
    out["gray"] = np.asarray(im.convert("L"))
    out["hflip"] = np.asarray(im.transpose(Image.FLIP_LEFT_RIGHT))
    np.savez_compressed(ROOT / "tests" / "golden" / "pil_ops.npz", **out)
    print("wrote pil_ops.npz with Pillow", Image.__version__ if hasattr(Image, "__version__") else "")


if __name__ == "__main__":
    main()
