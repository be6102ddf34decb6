"""Drop-in for nerf_qa.DISTS_pytorch.DISTS_pt_original.DISTS (the variant NeRFQAModel trains).

Differences from the canonical module (reference file DISTS_pt_original.py):
  * alpha/beta are clamped at load: alpha >= lb*ratio, beta >= lb (:65-72), and the published
    values are kept as original_alpha / original_beta (:67-68, read by model_stats.py:89);
  * forward applies optional relu / detach_beta / w_sum_detach from the run config (:111-119);
  * the result is `score.squeeze()` (:134): shape (B,) for B > 1 and 0-d for B == 1;
  * project_weights uses the configured lower bound on every channel (:88-95);
  * prepare_image(image, resize=True) always keeps the aspect ratio (:140-144; imported by
    run_nerf_qa.py:29 and nerf_qa/data_fr.py:34).
The VGG pyramid and the statistics are the same HIP kernels; only the 2950-term weighted sum is
evaluated in PyTorch so that autograd reaches alpha and beta.
"""
from __future__ import annotations

import numpy as np
import torch

from ..config import config
from .DISTS_pt import _DATA, DISTS as _BaseDISTS
from .DISTS_pt import L2pooling  # noqa: F401  (the reference defines it in this module too: pickles name it)
from .DISTS_pt import prepare_image as _prepare_image


class DISTS(_BaseDISTS):
    def __init__(self, load_weights=True, precision=None, vgg16_path=None):
        super().__init__(load_weights=False, precision=precision, vgg16_path=vgg16_path)
        if load_weights:
            ab = np.load(_DATA)
            alpha = torch.from_numpy(ab["alpha"]).view(1, -1, 1, 1).clone()
            beta = torch.from_numpy(ab["beta"]).view(1, -1, 1, 1).clone()
            self.original_alpha = alpha.clone()
            self.original_beta = beta.clone()
            lb = config().weight_lower_bound
            ab_ratio = config().alpha_beta_ratio
            self.alpha.data = torch.clamp(alpha, min=lb * ab_ratio)
            self.beta.data = torch.clamp(beta, min=lb)

    def _apply(self, fn, *args, **kwargs):  # keep original_alpha/beta on the module's device
        super()._apply(fn, *args, **kwargs)
        for name in ("original_alpha", "original_beta"):
            if hasattr(self, name):
                setattr(self, name, fn(getattr(self, name)))
        return self

    def project_weights(self):
        lower_bound = torch.zeros_like(self.alpha.data) + config().weight_lower_bound
        alpha = torch.max(self.alpha.data, lower_bound * config().alpha_beta_ratio)
        beta = torch.max(self.beta.data, lower_bound)
        weight_sum = torch.cat([alpha, beta], dim=1).sum()
        self.alpha.data = alpha / weight_sum
        self.beta.data = beta / weight_sum

    def _score_weights(self):  # (what `auto` re-walks its ladder with once alpha/beta have moved; DISTS_pt._live_choice)
        flags = str(config().dists_weight_norm).split("+")
        a, b = self.alpha.detach().reshape(-1).float(), self.beta.detach().reshape(-1).float()
        if "relu" in flags:
            a, b = torch.relu(a), torch.relu(b)
        wsum = a.sum() + b.sum()
        return a / wsum, b / wsum

    def forward(self, x, y, require_grad=False, batch_average=False):
        s1, s2 = self._similarities(x, y, require_grad)
        flags = str(config().dists_weight_norm).split("+")
        alpha = torch.relu(self.alpha) if "relu" in flags else self.alpha
        beta = torch.relu(self.beta) if "relu" in flags else self.beta
        if config().detach_beta == "True":
            beta = beta.detach()
        w_sum = alpha.sum() + beta.sum()
        if "w_sum_detach" in flags:
            w_sum = w_sum.detach()
        alpha, beta = alpha.view(1, -1) / w_sum, beta.view(1, -1) / w_sum
        dist1 = dist2 = 0
        o = 0
        for c in self.chns:
            dist1 = dist1 + (alpha[:, o:o + c] * s1[:, o:o + c]).sum(1, keepdim=True)
            dist2 = dist2 + (beta[:, o:o + c] * s2[:, o:o + c]).sum(1, keepdim=True)
            o += c
        score = 1 - (dist1 + dist2).squeeze()
        return score.mean() if batch_average else score


def prepare_image(image, resize=True):
    """PIL image -> float32 (1,3,H,W) in [0,1]; DISTS_pt_original.py:140-144: `resize(image, 256)` scales the
    short side to 256 and keeps the aspect ratio (long side = int(256 * long / short))."""
    return _prepare_image(image, resize=resize, keep_aspect_ratio=True)
