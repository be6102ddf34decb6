"""CPU oracle for the DISTS hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional fp32 restatement, on CPU PyTorch ops, of what the reference
computes in nerf_qa/DISTS_pytorch/DISTS_pt.py.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this package; the
shipped path (nerf_qa_amd) never does and fails loudly without its HIP library.

Parity status: PINNED against the reference itself.  oracle/make_goldens.py
imports the reference's own DISTS class in the authoring container (with a local
stand-in for the absent torchvision package and the weights of
nerf_qa_amd.synth), checks this restatement against it bit-for-bit and freezes
the outputs in tests/golden/.  The reference's own repository holds no tests or
golden vectors for this path, and the real ImageNet VGG
weights are unavailable offline, so parity on *those* weights is unpinned.

Every function cites the reference lines it restates.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

CHNS = (3, 64, 128, 256, 512, 512)           # DISTS_pt.py:57
STAGE_CONVS = (2, 2, 3, 3, 3)                # DISTS_pt.py:36-49
IMAGENET_MEAN = (0.485, 0.456, 0.406)        # DISTS_pt.py:54
IMAGENET_STD = (0.229, 0.224, 0.225)         # DISTS_pt.py:55


def hanning_filter() -> torch.Tensor:
    """3x3 L2-pool filter: outer(hanning(5)[1:-1]) normalised.  DISTS_pt.py:17-19."""
    a = np.hanning(5)[1:-1]
    g = torch.Tensor(a[:, None] * a[None, :])
    return g / torch.sum(g)


def l2pool(x: torch.Tensor) -> torch.Tensor:
    """sqrt(depthwise3x3_s2_p1(x^2) + 1e-12).  DISTS_pt.py:22-25 (= ADISTS.py:28-31)."""
    c = x.shape[1]
    filt = hanning_filter()[None, None].repeat(c, 1, 1, 1)
    out = F.conv2d(x ** 2, filt, stride=2, padding=1, groups=c)
    return (out + 1e-12).sqrt()


def vgg_pyramid(x: torch.Tensor, convs) -> list:
    """forward_once: the six tapped maps [x, relu1_2 ... relu5_3].  DISTS_pt.py:91-103.

    `convs` is a list of 13 (weight OIHW, bias) tensors in network order.  The raw
    image is tap 0; the network input is (x-mean)/std, zero-padded *after*
    normalisation; each later stage starts with an L2-pool.
    """
    mean = torch.tensor(IMAGENET_MEAN).view(1, -1, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, -1, 1, 1)
    h = (x - mean) / std
    feats = [x]
    li = 0
    for s, nconv in enumerate(STAGE_CONVS):
        if s > 0:
            h = l2pool(h)
        for _ in range(nconv):
            w, b = convs[li]
            h = F.relu(F.conv2d(h, w, b, stride=1, padding=1))
            li += 1
        feats.append(h)
    return feats


def dists_stats(feats0, feats1):
    """Per-channel similarities S1 (mean) and S2 (structure), each (B,1475).

    DISTS_pt.py:130-141 (same body at 190-201): population variance by two-pass,
    covariance by single-pass, c1 = c2 = 1e-6.
    """
    c1 = c2 = 1e-6
    s1, s2 = [], []
    for fx, fy in zip(feats0, feats1):
        xm = fx.mean([2, 3], keepdim=True)
        ym = fy.mean([2, 3], keepdim=True)
        s1.append(((2 * xm * ym + c1) / (xm ** 2 + ym ** 2 + c1)).flatten(1))
        xv = ((fx - xm) ** 2).mean([2, 3], keepdim=True)
        yv = ((fy - ym) ** 2).mean([2, 3], keepdim=True)
        cov = (fx * fy).mean([2, 3], keepdim=True) - xm * ym
        s2.append(((2 * cov + c2) / (xv + yv + c2)).flatten(1))
    return torch.cat(s1, 1), torch.cat(s2, 1)


def dists_score(s1, s2, alpha, beta, batch_average=False):
    """score_b = 1 - sum_c alpha_c/w * S1_bc - sum_c beta_c/w * S2_bc.  DISTS_pt.py:127-148.

    The reference sums stage by stage (six partial sums each for alpha and beta,
    then dist1+dist2); the same order is kept here so fp32 results agree bit for
    bit.
    """
    a = alpha.reshape(-1)
    b = beta.reshape(-1)
    w_sum = a.sum() + b.sum()
    an, bn = a / w_sum, b / w_sum
    d1 = torch.zeros(s1.shape[0])
    d2 = torch.zeros(s1.shape[0])
    o = 0
    for c in CHNS:
        d1 = d1 + (an[o:o + c] * s1[:, o:o + c]).sum(1)
        d2 = d2 + (bn[o:o + c] * s2[:, o:o + c]).sum(1)
        o += c
    score = 1 - (d1 + d2)
    return score.mean() if batch_average else score


def dists(x, y, convs, alpha, beta, batch_average=False):
    """DISTS.forward(x, y).  DISTS_pt.py:105-148."""
    with torch.no_grad():
        f0 = vgg_pyramid(x, convs)
        f1 = vgg_pyramid(y, convs)
        s1, s2 = dists_stats(f0, f1)
        return dists_score(s1, s2, alpha, beta, batch_average)


def project_weights(alpha, beta):
    """Clamp below (0.02 on the 3 image channels, 0 elsewhere) and renormalise.  DISTS_pt.py:82-89."""
    lb = torch.zeros_like(alpha)
    lb[:, :3] = 0.02
    a = torch.max(alpha, lb)
    b = torch.max(beta, lb)
    s = torch.cat([a, b], dim=1).sum()
    return a / s, b / s


def convs_from_numpy(np_convs):
    """synth.vgg16_weights() output -> list of torch (w, b)."""
    return [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in np_convs]
