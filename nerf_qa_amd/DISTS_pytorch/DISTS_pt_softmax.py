"""Drop-in for nerf_qa.DISTS_pytorch.DISTS_pt_softmax.DISTS.

alpha/beta are stored as logits, log(clamp(w, 0) + 1e-10) of the published weights
(DISTS_pt_softmax.py:70-78), and forward takes a softmax over the 2950 concatenated logits
(:117-121).  The reference then calls `.detach()` on a *tuple* when detach_beta == 'True'
(:122-123), which raises; here beta is detached per stage instead, which is what was meant.
"""
from __future__ import annotations

import numpy as np
import torch

from ..config import config
from .DISTS_pt import _DATA, DISTS as _BaseDISTS
from .DISTS_pt import L2pooling  # noqa: F401  (the reference defines it in this module too: pickles name it)


class DISTS(_BaseDISTS):
    def __init__(self, load_weights=True, precision=None, vgg16_path=None):
        super().__init__(load_weights=False, precision=precision, vgg16_path=vgg16_path)
        if load_weights:
            ab = np.load(_DATA)
            alpha = torch.from_numpy(ab["alpha"]).view(1, -1, 1, 1)
            beta = torch.from_numpy(ab["beta"]).view(1, -1, 1, 1)
            self.original_alpha, self.original_beta = alpha.clone(), beta.clone()
            logits = torch.log(torch.clamp(torch.cat([alpha, beta], dim=1), min=0.0) + 1e-10)
            a_logits, b_logits = torch.split(logits, [alpha.numel(), beta.numel()], dim=1)
            self.alpha.data = a_logits.clone()
            self.beta.data = b_logits.clone()

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        for name in ("original_alpha", "original_beta"):
            if hasattr(self, name):
                setattr(self, name, fn(getattr(self, name)))
        return self

    def _score_weights(self):  # (DISTS_pt._live_choice)
        w = torch.softmax(torch.cat([self.alpha.detach(), self.beta.detach()], dim=1).float(), dim=1).reshape(-1)
        return w[:self.alpha.numel()], w[self.alpha.numel():]

    def forward(self, x, y, require_grad=False, batch_average=False, warp=None, certainty=None):
        s1, s2 = self._similarities(x, y, require_grad)
        w = torch.softmax(torch.cat([self.alpha, self.beta], dim=1), dim=1)
        alpha, beta = torch.split(w, self.alpha.shape[1], dim=1)
        alpha, beta = alpha.view(1, -1), beta.view(1, -1)
        if config().detach_beta == "True":
            beta = beta.detach()
        dist1 = dist2 = 0
        o = 0
        for c in self.chns:
            dist1 = dist1 + (alpha[:, o:o + c] * s1[:, o:o + c]).sum(1, keepdim=True)
            dist2 = dist2 + (beta[:, o:o + c] * s2[:, o:o + c]).sum(1, keepdim=True)
            o += c
        score = 1 - (dist1 + dist2).squeeze()
        return score.mean() if batch_average else score
