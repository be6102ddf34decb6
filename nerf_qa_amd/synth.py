"""Deterministic synthetic data for the DISTS / A-DISTS hot path.

There is no network in the build or on the GPU box, so the real ImageNet VGG-16
checkpoint (`vgg16-397923af.pth`, fetched by the reference at
nerf_qa/DISTS_pytorch/DISTS_pt.py:30 and nerf_qa/ADISTS/ADISTS.py:38) is not
available.  Everything that needs numbers -- VGG weights, frame pairs -- is
produced here by a counter-based generator written in plain numpy integer
arithmetic, so the same values come out on every machine and in every process
(parity tests on the GPU box regenerate them instead of shipping 59 MB).

Nothing here is taken from the reference; it only has to be reproducible.
"""
from __future__ import annotations

import numpy as np

# VGG-16 "D" convolution plan: (cin, cout) per conv3x3, grouped by DISTS stage
# (stage boundaries follow DISTS_pt.py:36-49: conv indices 0,2 | 5,7 | 10,12,14 |
# 17,19,21 | 24,26,28 of torchvision's `features`).
VGG_STAGES = (
    ((3, 64), (64, 64)),
    ((64, 128), (128, 128)),
    ((128, 256), (256, 256), (256, 256)),
    ((256, 512), (512, 512), (512, 512)),
    ((512, 512), (512, 512), (512, 512)),
)
VGG_CONVS = tuple(c for s in VGG_STAGES for c in s)
# index of each conv inside torchvision's vgg16().features Sequential
VGG_FEATURE_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
CHNS = (3, 64, 128, 256, 512, 512)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(seed: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        k = _mix(np.array([(seed * 0x100000001B3 + stream * 0x9E3779B1 + 0x51ED27) & 0xFFFFFFFFFFFFFFFF],
                          dtype=np.uint64))
        k = _mix(k)
    return k[0]


def uniform(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = _mix(_mix(idx ^ _key(seed, stream)))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 standard normals (Box-Muller over two uniform streams)."""
    u1 = uniform(seed, n, 2 * stream + 101)
    u2 = uniform(seed, n, 2 * stream + 102)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * np.pi * u2)


def vgg16_weights(seed: int = 1234, gain: float = 1.0):
    """List of 13 (weight OIHW float32, bias float32) numpy pairs.

    Fan-in He scaling, sqrt(2/(9*cin)), keeps post-ReLU activations O(1) through
    all five stages so the 16-bit feature path and the L2-pool squaring see a
    realistic dynamic range; `gain` scales every layer (gain>1 grows activations
    with depth, as the ImageNet weights do).  Biases are small and signed.
    """
    out = []
    for li, (cin, cout) in enumerate(VGG_CONVS):
        n = cout * cin * 9
        w = normal(seed, n, stream=2 * li) * (gain * np.sqrt(2.0 / (9.0 * cin)))
        b = normal(seed, cout, stream=2 * li + 1) * 0.05 + 0.02
        out.append((w.astype(np.float32).reshape(cout, cin, 3, 3), b.astype(np.float32)))
    return out


def _box5(x: np.ndarray) -> np.ndarray:
    """5x5 box blur with edge replication, NCHW float64."""
    p = np.pad(x, ((0, 0), (0, 0), (2, 2), (2, 2)), mode="edge")
    acc = np.zeros_like(x)
    h, w = x.shape[2], x.shape[3]
    for dy in range(5):
        for dx in range(5):
            acc += p[:, :, dy:dy + h, dx:dx + w]
    return acc / 25.0


KINDS = ("noise02", "noise10", "blur", "indep", "same")
# NeRF-render-like content (round 4): constant backgrounds, smooth regions, flat frames with floaters -- the regime
# that produces nearly dead VGG channels (what the 16-bit modes' outliers and A-DISTS' knife edge are made of), which
# the full-frame-texture families above never reach (a reference script feeds exactly such frames: prep.py:181-198)
NERF_KINDS = ("nerf_white", "nerf_black", "nerf_grad", "nerf_float")


def _object_mask(seed: int, h: int, w: int, frac_lo=0.3, frac_hi=0.6) -> np.ndarray:
    """A soft-edged union of two ellipses covering frac_lo..frac_hi of the frame ((1,1,h,w) float64 in [0,1]): the
    'object' of a synthetic NeRF scene; everything else is constant background."""
    u = uniform(seed, 12, stream=21)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64) / h, np.arange(w, dtype=np.float64) / w, indexing="ij")
    frac = frac_lo + (frac_hi - frac_lo) * u[0]
    m = np.zeros((h, w))
    for k in range(2):
        cy, cx = 0.35 + 0.3 * u[1 + 5 * k], 0.35 + 0.3 * u[2 + 5 * k]
        ry = np.sqrt(frac / 2.0 / np.pi) * (0.8 + 0.6 * u[3 + 5 * k])
        rx = frac / 2.0 / np.pi / ry
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        m = np.maximum(m, np.clip((1.0 - d) * 12.0, 0.0, 1.0))  # ~2-3 pixels of anti-aliased rim at 256 x 256
    return m[None, None]


def _nerf_pair(seed: int, h: int, w: int, kind: str):
    n = 3 * h * w
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    ph = uniform(seed, 6, stream=8) * 2.0 * np.pi
    low = np.stack([0.5 + 0.5 * np.sin(xx * (0.05 + 0.01 * c) + ph[c]) * np.cos(yy * (0.04 + 0.013 * c) + ph[3 + c])
                    for c in range(3)])[None]
    tex = 0.5 * uniform(seed, n, stream=7).reshape(1, 3, h, w) + 0.5 * low
    if kind in ("nerf_white", "nerf_black"):
        bg = 1.0 if kind == "nerf_white" else 0.0
        m = _object_mask(seed, h, w)
        x = m * tex + (1.0 - m) * bg
        # the render: the object slightly blurred and noisy, the background EXACTLY the constant (as a NeRF's
        # white / black background is), the silhouette a little off (mask of another seed-derived shape blended in)
        m2 = np.clip(m + 0.15 * (_object_mask(seed + 7919, h, w, 0.3, 0.6) - m), 0.0, 1.0)
        obj = _box5(tex) * 0.5 + tex * 0.5 + 0.03 * normal(seed, n, stream=9).reshape(tex.shape)
        y = np.clip(m2 * obj + (1.0 - m2) * bg, 0.0, 1.0)
    elif kind == "nerf_grad":
        # smooth everywhere: a low-frequency pattern, no per-pixel noise; the render is brighter / softer
        u = uniform(seed, 2, stream=22)
        x = low
        y = np.clip(_box5(low) * (0.9 + 0.2 * u[0]) + 0.04 * (u[1] - 0.5), 0.0, 1.0)
    elif kind == "nerf_float":
        # a flat frame (one colour, a faint gradient) and a render with a handful of small blobs ('floaters')
        u = uniform(seed, 64, stream=23)
        col = 0.2 + 0.6 * u[:3]
        x = col[None, :, None, None] + 0.05 * (low - 0.5)
        y = x.copy()
        for k in range(6):
            cy, cx, r = u[4 + 6 * k] * h, u[5 + 6 * k] * w, 2.0 + u[6 + 6 * k] * 0.04 * min(h, w)
            blob = np.clip(1.5 - np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2) / r, 0.0, 1.0)[None, None]
            y = (1.0 - blob) * y + blob * u[7 + 6 * k:10 + 6 * k][None, :, None, None]
        x, y = np.clip(x, 0.0, 1.0), np.clip(y, 0.0, 1.0)
    else:
        raise ValueError(f"unknown distortion kind {kind!r}")
    return x.astype(np.float32), y.astype(np.float32)


def frame_pair(seed: int, h: int, w: int, kind: str = "noise10"):
    """One synthetic (x, y) pair, each float32 (1,3,h,w) in [0,1].

    x is a smooth-ish random image (uniform noise mixed with a low-frequency
    pattern so that blur actually changes structure); y is a distortion of x:
    additive Gaussian noise (sigma .02 / .1), a 5x5 box blur, an independent
    image, or x itself.
    """
    if kind in NERF_KINDS:
        return _nerf_pair(seed, h, w, kind)
    n = 3 * h * w
    base = uniform(seed, n, stream=7).reshape(1, 3, h, w)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    ph = uniform(seed, 6, stream=8) * 2.0 * np.pi
    low = np.stack([0.5 + 0.5 * np.sin(xx * (0.05 + 0.01 * c) + ph[c]) * np.cos(yy * (0.04 + 0.013 * c) + ph[3 + c])
                    for c in range(3)])[None]
    x = 0.6 * base + 0.4 * low
    if kind == "noise02":
        y = np.clip(x + 0.02 * normal(seed, n, stream=9).reshape(x.shape), 0.0, 1.0)
    elif kind == "noise10":
        y = np.clip(x + 0.10 * normal(seed, n, stream=9).reshape(x.shape), 0.0, 1.0)
    elif kind == "blur":
        y = _box5(x)
    elif kind == "indep":
        y = uniform(seed, n, stream=10).reshape(x.shape)
    elif kind == "same":
        y = x.copy()
    else:
        raise ValueError(f"unknown distortion kind {kind!r}")
    return x.astype(np.float32), y.astype(np.float32)


def frame_batch(seeds, h: int, w: int, kinds=None):
    """Stack frame_pair over seeds; kinds cycles through the 4 distortions."""
    xs, ys = [], []
    for i, s in enumerate(seeds):
        kind = (kinds[i % len(kinds)] if kinds else KINDS[i % 4])
        x, y = frame_pair(int(s), h, w, kind)
        xs.append(x)
        ys.append(y)
    return np.concatenate(xs, 0), np.concatenate(ys, 0)
