"""Where the frozen VGG-16 feature weights come from.

The reference fetches torchvision's ImageNet checkpoint over the network
(`models.vgg16(pretrained=True)`, nerf_qa/DISTS_pytorch/DISTS_pt.py:30).  Offline that is
impossible, so:
  * if NQA_VGG16_WEIGHTS (or the `vgg16_path` argument) names a local torchvision state
    dict (`vgg16-397923af.pth`, keys `features.N.weight/bias`), it is loaded;
  * otherwise the deterministic stand-in weights of nerf_qa_amd.synth are used and a
    warning says that scores are then NOT comparable with published DISTS values.
"""
from __future__ import annotations

import os
import warnings

import torch

from . import synth


def load_vgg16_convs(vgg16_path: str | None = None, seed: int = 1234):
    """-> list of 13 (weight OIHW float32 tensor, bias float32 tensor), and a source tag."""
    path = vgg16_path or os.environ.get("NQA_VGG16_WEIGHTS")
    if path:
        sd = torch.load(path, map_location="cpu")
        if "state_dict" in sd:
            sd = sd["state_dict"]
        convs = []
        for idx, (cin, cout) in zip(synth.VGG_FEATURE_IDX, synth.VGG_CONVS):
            w = sd[f"features.{idx}.weight"].float().contiguous()
            b = sd[f"features.{idx}.bias"].float().contiguous()
            if tuple(w.shape) != (cout, cin, 3, 3):
                raise ValueError(f"{path}: features.{idx}.weight has shape {tuple(w.shape)}")
            convs.append((w, b))
        return convs, f"file:{path}"
    warnings.warn("nerf_qa_amd: no VGG-16 checkpoint given (NQA_VGG16_WEIGHTS); using deterministic stand-in "
                  "weights -- scores are self-consistent but not comparable with published DISTS numbers",
                  stacklevel=3)
    return [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in synth.vgg16_weights(seed)], f"synth:{seed}"
