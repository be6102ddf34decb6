"""Drop-in for nerf_qa.model_stats.NeRFQAModel: DISTS + a linear / sqrt / 4-parameter logistic
head fitted on the training table (model_stats.py:23-102).  Host-side PyTorch, as north_star
prescribes for the regression head; the DISTS scores come from the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .config import config


class NeRFQAModel(nn.Module):
    def __init__(self, train_df, precision=None, vgg16_path=None):
        super().__init__()
        cfg = config()
        X = np.asarray(train_df["DISTS"].values, dtype=np.float64)
        y = np.asarray(train_df[cfg.subjective_score_type].values, dtype=np.float64)
        if cfg.regression_type == "logistic":
            from scipy.optimize import curve_fit
            sign = 1.0 if cfg.subjective_score_type == "MOS" else -1.0

            def logistic(x, beta1, beta2, beta3, beta4):
                return (beta1 - beta2) / (1 + np.exp(sign * (x - beta3) / np.abs(beta4))) + beta2

            mos = cfg.subjective_score_type == "MOS"
            p0 = [np.max(y) if mos else np.min(y), np.min(y) if mos else np.max(y), np.median(X), np.std(X)]
            params, _ = curve_fit(logistic, X, y, p0=p0)
            self.b1, self.b2, self.b3, self.b4 = (nn.Parameter(torch.tensor([p], dtype=torch.float32)) for p in params)
        else:
            A = np.sqrt(X) if cfg.regression_type == "sqrt" else X
            coef, intercept = np.polyfit(A, y, 1)  # ordinary least squares, = sklearn LinearRegression
            self.dists_weight = nn.Parameter(torch.tensor([coef], dtype=torch.float32))
            self.dists_bias = nn.Parameter(torch.tensor([intercept], dtype=torch.float32))
        if cfg.dists_weight_norm == "softmax":
            from .DISTS_pytorch.DISTS_pt_softmax import DISTS
        else:
            from .DISTS_pytorch.DISTS_pt_original import DISTS
        self.dists_model = DISTS(precision=precision, vgg16_path=vgg16_path)

    def logistic(self, dists_scores):
        sign = 1.0 if config().subjective_score_type == "MOS" else -1.0
        return (self.b1 - self.b2) / (1 + torch.exp(sign * (dists_scores - self.b3) / torch.abs(self.b4))) + self.b2

    def sqrt(self, dists_scores):
        return torch.sqrt(dists_scores) * self.dists_weight + self.dists_bias

    def linear(self, dists_scores):
        return dists_scores * self.dists_weight + self.dists_bias

    def entropy_loss(self):
        weights = torch.cat([self.dists_model.alpha, self.dists_model.beta], dim=1)
        if config().dists_weight_norm == "softmax":
            weights = torch.softmax(weights, dim=1)
        else:
            if config().dists_weight_norm == "relu":
                weights = torch.relu(weights)
            weights = weights / weights.sum()
        original = torch.cat([self.dists_model.original_alpha, self.dists_model.original_beta], dim=1)
        return -torch.sum(original * torch.log(weights + 1e-10))

    def forward(self, dist, ref):
        dists_scores = self.dists_model(dist, ref)
        kind = config().regression_type
        if kind == "logistic":
            scores = self.logistic(dists_scores)
        elif kind == "sqrt":
            scores = self.sqrt(dists_scores)
        else:
            scores = self.linear(dists_scores)
        return scores, dists_scores
