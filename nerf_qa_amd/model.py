"""Drop-in for nerf_qa.model.NeRFQAModel (nerf_qa/model.py:22-56; imported by run.py:26,
run_test2.py:27, run_test2_cross.py:29, run_test2_sf.py:29): DISTS + a linear head fitted on the
training table, selected by `wandb.config.mode`:
  mode contains "sqrt"     the head is fitted on, and applied to, sqrt(DISTS)        (:26-27,52-53)
  mode contains "softmax"  the DISTS variant is DISTS_pt_softmax, else _original     (:40-43)
The response column is always 'MOS' (:30).  Host-side PyTorch, as north_star prescribes for the
regression head; the DISTS scores come from the HIP path.  Only the FR model is mirrored: the
NR models of the same reference file (model.py:60-163) need torch.hub downloads and are out of
scope.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .config import config


class NeRFQAModel(nn.Module):
    def __init__(self, train_df, precision=None, vgg16_path=None):
        super().__init__()
        mode = config().mode
        X = np.asarray(train_df["DISTS"].values, dtype=np.float64)
        if mode in ("sqrt", "softmax+sqrt"):
            X = np.sqrt(X)
        y = np.asarray(train_df["MOS"].values, dtype=np.float64)
        coef, intercept = np.polyfit(X, y, 1)  # ordinary least squares, = sklearn LinearRegression (:33-34)
        if mode in ("softmax", "softmax+sqrt"):
            from .DISTS_pytorch.DISTS_pt_softmax import DISTS
        else:
            from .DISTS_pytorch.DISTS_pt_original import DISTS
        self.dists_model = DISTS(precision=precision, vgg16_path=vgg16_path)
        self.dists_weight = nn.Parameter(torch.tensor([coef], dtype=torch.float32))
        self.dists_bias = nn.Parameter(torch.tensor([intercept], dtype=torch.float32))

    def forward(self, dist, ref):
        dists_scores = self.dists_model(dist, ref)
        if config().mode in ("sqrt", "softmax+sqrt"):
            scores = torch.sqrt(dists_scores) * self.dists_weight + self.dists_bias
        else:
            scores = dists_scores * self.dists_weight + self.dists_bias
        return scores, dists_scores
