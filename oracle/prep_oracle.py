"""CPU oracle for the input-preparation row  --  TEST INFRASTRUCTURE.

Only tests/ import this.  Three pieces:
  to_tensor / to_tensor_roundtrip   the reference's own torch expressions (prep.py:89-91, data.py:80);
  interp                            F.interpolate exactly as the reference calls it (prep.py:93-95);
  pil_resize_bilinear_u8            numpy restatement of Pillow's 8-bit ImagingResample with the
                                    BILINEAR filter (third-party: poetry.lock pins pillow 10.2.0;
                                    restated from its published algorithm, src/libImaging/Resample.c:
                                    precompute_coeffs, normalize_coeffs_8bpc,
                                    ImagingResampleHorizontal_8bpc / Vertical_8bpc).  Pinned in
                                    tests/test_prep_oracle.py against the Pillow installed in this
                                    image (Image.resize(..., BILINEAR)), bit for bit.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

PRECISION_BITS = 32 - 8 - 2


def to_tensor(frames_u8: np.ndarray) -> torch.Tensor:
    """(n,H,W,3) uint8 -> (n,3,H,W) float32 / 255.  prep.py:89, data.py:80."""
    return torch.from_numpy(frames_u8).permute(0, 3, 1, 2).float() / 255.0


def to_tensor_roundtrip(frames_u8: np.ndarray) -> torch.Tensor:
    """prep.py:89-91: /255 -> ToPILImage (mul(255).byte()) -> ToTensor (/255)."""
    f = to_tensor(frames_u8)
    return f.mul(255).byte().to(torch.float32).div(255)


def interp(x: torch.Tensor, size) -> torch.Tensor:
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def _coeffs(in_size: int, out_size: int):
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One resampling pass along `axis` (0 = rows, 1 = columns) of an (H,W,3) uint8 image."""
    bounds, kk = _coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for o in range(out_size):
        lo, cnt = bounds[o]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[o, :cnt], src[lo:lo + cnt], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img: np.ndarray, size) -> np.ndarray:
    """Image.fromarray(img).resize((W,H), BILINEAR) for an (Hin,Win,3) uint8 array; size = (H, W)."""
    ho, wo = int(size[0]), int(size[1])
    if wo != img.shape[1]:
        img = _pass(img, wo, 1)
    if ho != img.shape[0]:
        img = _pass(img, ho, 0)
    return img
