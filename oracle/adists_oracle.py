"""CPU oracle for the A-DISTS hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional fp32 restatement of nerf_qa/ADISTS/ADISTS.py on CPU PyTorch ops.
Same rules as dists_oracle.py: imported only by tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke(); pinned against the imported reference by
oracle/make_goldens.py (tests/golden/adists_*.npz).

The reference decides between windowed and global statistics with a bare
try/except around a *valid* 21x21 convolution (ADISTS.py:78-97,168-180); the
exception fires exactly when the map is smaller than the window in either
dimension, which is what `windowed()` tests here.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .dists_oracle import CHNS, vgg_pyramid  # the pyramid is the same network (ADISTS.py:112-125)

WINDOW = 21


def gaussian_1d(window_size: int = WINDOW, sigma: float = WINDOW / 3) -> torch.Tensor:
    """ADISTS.py:102-104 (sigma = window_size/3, ADISTS.py:69)."""
    g = torch.Tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])
    return g / g.sum()


def window_2d(channels: int, window_size: int = WINDOW) -> torch.Tensor:
    """Outer-product Gaussian window expanded per channel.  ADISTS.py:106-110."""
    g = gaussian_1d(window_size, window_size / 3).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channels, 1, window_size, window_size).contiguous()


def windowed(h: int, w: int, window_size: int = WINDOW) -> bool:
    return h >= window_size and w >= window_size


def _wconv(t, win):
    return F.conv2d(t, win, stride=1, padding=0, groups=t.shape[1])


def _minmax(p, c0):
    lo, _ = p.flatten(2).min(dim=-1, keepdim=True)
    hi, _ = p.flatten(2).max(dim=-1, keepdim=True)
    return (p - lo.unsqueeze(-1)) / (hi.unsqueeze(-1) - lo.unsqueeze(-1) + c0)


def compute_prob(feats, window_size: int = WINDOW):
    """Texture-probability maps, coarse to fine, from the x pyramid only.  ADISTS.py:71-100."""
    c0 = 1e-12
    x = feats[0]
    ps_prod = torch.ones_like(x[:, 0:1])
    out = []
    for k in range(len(feats) - 1, -1, -1):
        f = feats[k]
        if windowed(f.shape[2], f.shape[3], window_size):
            win = window_2d(f.shape[1], window_size)
            m = _wconv(f, win)
            v = _wconv(f ** 2, win) - m ** 2
            h, w = m.shape[2], m.shape[3]
            gamma = torch.mean(v / (m + c0), dim=1, keepdim=True)
            z = (gamma - gamma.mean(dim=(2, 3), keepdim=True)) / (gamma.std(dim=(2, 3), keepdim=True) + c0)
            ps = _minmax(1 / (1 + torch.exp(-z)), c0)
            ps_prod = ps * F.interpolate(ps_prod, size=(h, w), mode="bilinear", align_corners=True)
            ps_prod = _minmax(ps_prod, c0)
        else:
            m = f.mean([2, 3], keepdim=True)
            v = ((f - m) ** 2).mean([2, 3], keepdim=True)
            gamma = torch.mean(v / (m + c0), dim=1, keepdim=True)
            ps = 1 / (1 + torch.exp(-gamma))
            ps_prod = ps * F.interpolate(ps_prod, size=(1, 1), mode="bilinear", align_corners=True)
        out.append(ps_prod)
    return out[::-1]


def entropy_weight(feat):
    """Per-channel spatial entropy, scaled by C.  ADISTS.py:127-135."""
    c0 = 1e-12
    b, c, h, w = feat.shape
    p = F.normalize(F.relu(feat), dim=(2, 3)).reshape(b, c, -1)
    p = p / (torch.sum(p, dim=2, keepdim=True) + c0)
    wgt = torch.sum(-p * torch.log2(p + c0), dim=2, keepdim=True)
    wgt = wgt / (wgt.sum(dim=1, keepdim=True) + c0)
    return wgt * c


def channel_weights(feats_x):
    """Concatenate, normalise, clamp to mean +- 0.5 std (population), renormalise.  ADISTS.py:150-161."""
    wgt = torch.concat([entropy_weight(f) for f in feats_x], dim=1)
    wgt = wgt / wgt.sum(dim=(1, 2), keepdim=True)
    mu = wgt.mean(dim=(1, 2), keepdim=True)
    sd = torch.sqrt(((wgt - mu) ** 2).mean(dim=(1, 2), keepdim=True))
    wgt = wgt.clamp(min=mu - 0.5 * sd, max=mu + 0.5 * sd)
    wgt = wgt / wgt.sum(dim=(1, 2), keepdim=True)
    return torch.split(wgt, list(CHNS), dim=1)


def adists_from_feats(feats_x, feats_y, window_size: int = WINDOW, as_loss: bool = False, as_map: bool = False):
    """ADISTS.forward after the two pyramids: (B,) scores, or 1-mean(D) if as_loss.  ADISTS.py:147-197.

    as_map=True returns the reference's (B,B,H,W) tensor: D_map_full starts as zeros (B,H,W) (:163)
    and every stage adds a (B,1,H,W) resized map (:189), so the sum broadcasts to out[i,j] = map[i]."""
    ps_x = compute_prob(feats_x, window_size)
    wl = channel_weights(feats_x)
    d = 0
    bsz, _, big_h, big_w = feats_x[0].shape
    d_map_full = torch.zeros([bsz, big_h, big_w])
    for k in range(len(CHNS) - 1, -1, -1):
        fx = F.normalize(feats_x[k], dim=(2, 3))
        fy = F.normalize(feats_y[k], dim=(2, 3))
        if windowed(fx.shape[2], fx.shape[3], window_size):
            win = window_2d(CHNS[k], window_size)
            xm = _wconv(fx, win)
            ym = _wconv(fy, win)
            xv = _wconv(fx ** 2, win) - xm ** 2
            yv = _wconv(fy ** 2, win) - ym ** 2
            cov = _wconv(fx * fy, win) - xm * ym
        else:
            xm = fx.mean([2, 3], keepdim=True)
            ym = fy.mean([2, 3], keepdim=True)
            xv = ((fx - xm) ** 2).mean([2, 3], keepdim=True)
            yv = ((fy - ym) ** 2).mean([2, 3], keepdim=True)
            cov = (fx * fy).mean([2, 3], keepdim=True) - xm * ym
        t = (2 * xm * ym + 1e-6) / (xm ** 2 + ym ** 2 + 1e-6)
        s = (2 * cov + 1e-6) / (xv + yv + 1e-6)
        ps = ps_x[k].expand(xm.shape[0], xm.shape[1], -1, -1)
        pt = 1 - ps
        d_map = ((pt * t + ps * s) * wl[k].unsqueeze(3)).sum(1, keepdim=True)
        if as_map:
            d_map_full = d_map_full + F.interpolate(d_map, size=(big_h, big_w), mode="bilinear", align_corners=False)
        d = d + d_map.mean([2, 3]).sum(1)
    if as_map:
        return 1 - d_map_full
    return 1 - d.mean() if as_loss else 1 - d


def adists(x, y, convs, as_loss=False, window_size: int = WINDOW, as_map: bool = False):
    """ADISTS.forward(x, y, as_loss, as_map).  ADISTS.py:137-197.  x drives ps and weights."""
    assert x.shape == y.shape
    with torch.no_grad():
        fx = vgg_pyramid(x, convs)
        fy = vgg_pyramid(y, convs)
        return adists_from_feats(fx, fy, window_size, as_loss, as_map)
