"""Where the frozen VGG-16 feature weights come from.

The reference fetches torchvision's ImageNet checkpoint over the network
(`models.vgg16(pretrained=True)`, nerf_qa/DISTS_pytorch/DISTS_pt.py:30).  Offline that is
impossible, so the source has to be named -- by the `vgg16_path` argument of DISTS / ADISTS or by
the NQA_VGG16_WEIGHTS environment variable:
  * a path to a local torchvision state dict (`vgg16-397923af.pth`, keys
    `features.N.weight/bias`): loaded with `weights_only=True`;
  * "synth", "synth:<seed>" or "synth:<seed>:<gain>": the deterministic stand-in weights of
    nerf_qa_amd.synth (what tests, bench.py and smoke() ask for).  Scores are then
    self-consistent but NOT comparable with published DISTS values, and every precision bar in
    README / DESIGN was measured on these stand-ins only.
With neither, construction raises: a drop-in user must never get plausible-looking scores from
made-up weights without having asked for them.
"""
from __future__ import annotations

import os

import torch

from . import synth
from ._lib import NqaError


def parse_synth(spec: str):
    """"synth[:seed[:gain]]" -> (seed, gain) or None if `spec` is not a synth request."""
    if spec != "synth" and not spec.startswith("synth:"):
        return None
    parts = spec.split(":")
    if len(parts) > 3:
        raise ValueError(f"bad synthetic-weights spec {spec!r}; use synth[:seed[:gain]]")
    seed = int(parts[1]) if len(parts) > 1 and parts[1] else 1234
    gain = float(parts[2]) if len(parts) > 2 and parts[2] else 1.0
    return seed, gain


def load_vgg16_convs(vgg16_path: str | None = None):
    """-> list of 13 (weight OIHW float32 tensor, bias float32 tensor), and a source tag."""
    path = vgg16_path or os.environ.get("NQA_VGG16_WEIGHTS")
    if not path:
        raise NqaError("no VGG-16 weights named: pass vgg16_path= (or set NQA_VGG16_WEIGHTS) to a local torchvision "
                       "vgg16 state dict (vgg16-397923af.pth; the reference downloads it, DISTS_pt.py:30), or ask "
                       "for the deterministic stand-in explicitly with 'synth[:seed[:gain]]'")
    synth_spec = parse_synth(path)
    if synth_spec is not None:
        seed, gain = synth_spec
        tag = f"synth:{seed}" if gain == 1.0 else f"synth:{seed}:{gain:g}"
        return [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in synth.vgg16_weights(seed, gain)], tag
    sd = torch.load(path, map_location="cpu", weights_only=True)
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
