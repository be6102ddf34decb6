"""The A-DISTS head in differentiable torch operations: what `ADISTS.forward(as_loss=True)` needs under autograd.

nerf_qa/ADISTS/ADISTS.py:139-141 runs both pyramids WITH autograd when `as_loss=True`, so the loss carries gradients
through the texture probabilities (:71-100, from x only), the entropy channel weights (:127-135, :149-161, from x only)
and the windowed T / S terms (:165-191) back to both images.  The scoring path (`as_loss=False`, no gradient: every
caller in the reference) is the fused HIP kernel `nqa_adists_forward`; THIS file is used only when a gradient is actually
asked for: the tapped maps come from `autograd.PyramidTaps` (HIP forward, HIP backward through the 13 conv layers) and the
head below is plain torch on the GPU, so autograd differentiates it.  It is written for that purpose -- the 21 x 21
Gaussian window as two 1-D passes, the fall-back to global moments where a map is smaller than the window (the
reference's try / except) decided from the shape -- and is checked against float64 autograd over the CPU oracle
(tests/test_gpu_backward.py)."""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn.functional as F

CHNS = (3, 64, 128, 256, 512, 512)
C0 = 1e-12  # ADISTS.py:76,128
EPS = 1e-6  # ADISTS.py:182-183


def gauss_1d(window_size: int, like: torch.Tensor) -> torch.Tensor:
    """Normalised 1-D Gaussian of sigma = window_size / 3 (ADISTS.py:69,102-104); the 2-D window is its outer product."""
    sigma = window_size / 3
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)],
                     dtype=torch.float32)
    return (g / g.sum()).to(device=like.device, dtype=like.dtype)


def _window_mean(t: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """Valid depthwise correlation of (B,C,H,W) with the outer-product window g g^T, as a column pass and a row pass of
    shifted slices (n multiply-adds each).  Not F.conv2d: the texture probabilities divide by window means that are 0 or
    nearly 0 on dead regions of a channel, which puts upstream gradients of the order of 1 / 1e-12 next to ordinary ones;
    a sum of shifted slices (and its autograd transpose) never lets two windows' terms meet, whereas the library
    convolution's backward measured 24 % rms away from the CPU's on relu1_2 (tools/gpu_adists_grad_diag.py)."""
    n = g.numel()
    h, w = t.shape[2] - n + 1, t.shape[3] - n + 1
    col = g[0] * t[:, :, 0:h, :]
    for i in range(1, n):
        col = col + g[i] * t[:, :, i:i + h, :]
    out = g[0] * col[:, :, :, 0:w]
    for i in range(1, n):
        out = out + g[i] * col[:, :, :, i:i + w]
    return out


def _windowed(t: torch.Tensor, window_size: int) -> bool:
    return t.shape[2] >= window_size and t.shape[3] >= window_size  # (where F.conv2d would raise: ADISTS.py:79,91)


def _minmax(p: torch.Tensor) -> torch.Tensor:
    lo = p.flatten(2).amin(dim=-1, keepdim=True).unsqueeze(-1)
    hi = p.flatten(2).amax(dim=-1, keepdim=True).unsqueeze(-1)
    return (p - lo) / (hi - lo + C0)


def texture_probabilities(feats: Sequence[torch.Tensor], window_size: int) -> List[torch.Tensor]:
    """ps of every stage, coarse to fine products (ADISTS.py:71-100); feats = [image, relu1_2 .. relu5_3] of x."""
    prod = torch.ones_like(feats[0][:, 0:1])
    out = []
    for f in reversed(feats):
        if _windowed(f, window_size):
            g = gauss_1d(window_size, f)
            m = _window_mean(f, g)
            v = _window_mean(f * f, g) - m * m
            gamma = (v / (m + C0)).mean(dim=1, keepdim=True)
            z = (gamma - gamma.mean(dim=(2, 3), keepdim=True)) / (gamma.std(dim=(2, 3), keepdim=True) + C0)
            ps = _minmax(torch.sigmoid(z))
            prod = _minmax(ps * F.interpolate(prod, size=ps.shape[2:], mode="bilinear", align_corners=True))
        else:
            m = f.mean(dim=(2, 3), keepdim=True)
            v = ((f - m) ** 2).mean(dim=(2, 3), keepdim=True)
            gamma = (v / (m + C0)).mean(dim=1, keepdim=True)
            prod = torch.sigmoid(gamma) * F.interpolate(prod, size=(1, 1), mode="bilinear", align_corners=True)
        out.append(prod)
    return out[::-1]


def channel_weights(feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Entropy weights of all 1 475 channels, clamped to mean +- std / 2 and renormalised (ADISTS.py:127-135, :149-161)."""
    per = []
    for f in feats:
        b, c = f.shape[:2]
        p = F.normalize(F.relu(f), dim=(2, 3)).reshape(b, c, -1)
        p = p / (p.sum(dim=2, keepdim=True) + C0)
        h = -(p * torch.log2(p + C0)).sum(dim=2, keepdim=True)
        per.append(h / (h.sum(dim=1, keepdim=True) + C0) * c)
    w = torch.cat(per, dim=1)
    w = w / w.sum(dim=(1, 2), keepdim=True)
    mu = w.mean(dim=(1, 2), keepdim=True)
    sd = ((w - mu) ** 2).mean(dim=(1, 2), keepdim=True).sqrt()
    w = torch.maximum(torch.minimum(w, mu + 0.5 * sd), mu - 0.5 * sd)
    w = w / w.sum(dim=(1, 2), keepdim=True)
    return list(torch.split(w, [f.shape[1] for f in feats], dim=1))


def adists_d(feats_x: Sequence[torch.Tensor], feats_y: Sequence[torch.Tensor], window_size: int = 21) -> torch.Tensor:
    """D (B,) of ADISTS.py:147-191 from the two pyramids (lists of six NCHW maps); the module returns 1 - D or 1 - mean(D)."""
    ps_x = texture_probabilities(feats_x, window_size)
    wl = channel_weights(feats_x)
    d = 0
    for k in range(len(feats_x) - 1, -1, -1):
        fx, fy = F.normalize(feats_x[k], dim=(2, 3)), F.normalize(feats_y[k], dim=(2, 3))
        if _windowed(fx, window_size):
            g = gauss_1d(window_size, fx)
            xm, ym = _window_mean(fx, g), _window_mean(fy, g)
            xv = _window_mean(fx * fx, g) - xm * xm
            yv = _window_mean(fy * fy, g) - ym * ym
            cov = _window_mean(fx * fy, g) - xm * ym
        else:
            xm, ym = fx.mean(dim=(2, 3), keepdim=True), fy.mean(dim=(2, 3), keepdim=True)
            xv = ((fx - xm) ** 2).mean(dim=(2, 3), keepdim=True)
            yv = ((fy - ym) ** 2).mean(dim=(2, 3), keepdim=True)
            cov = (fx * fy).mean(dim=(2, 3), keepdim=True) - xm * ym
        t = (2 * xm * ym + EPS) / (xm * xm + ym * ym + EPS)
        s = (2 * cov + EPS) / (xv + yv + EPS)
        ps = ps_x[k]
        d_map = (((1 - ps) * t + ps * s) * wl[k].unsqueeze(3)).sum(dim=1, keepdim=True)
        d = d + d_map.mean(dim=(2, 3)).sum(dim=1)
    return d
