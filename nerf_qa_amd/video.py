"""Per-video scoring harness: the build's counterpart of the reference's offline loops
(prep.py:181-216, test2_prep.py:146-193): frames -> batched DISTS (+ A-DISTS) -> per-video
mean / std / min / max columns.  Host-side glue only; the scores come from the HIP modules.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import prep, sharding


def video_columns(name: str, frame_scores: np.ndarray) -> Dict[str, float]:
    """The four per-video columns the reference writes for a metric (prep.py:191-216)."""
    s = np.asarray(frame_scores, dtype=np.float64)
    return {name: float(np.mean(s)), f"{name}_std": float(np.std(s)), f"{name}_min": float(np.min(s)),
            f"{name}_max": float(np.max(s))}


def format_frame_scores(frame_scores: np.ndarray) -> str:
    """Frame-score list as the reference serialises it ('{:.6e}', test2_prep.py:123-125)."""
    return "[" + ", ".join("{:.6e}".format(float(v)) for v in frame_scores) + "]"


@torch.no_grad()
def score_video(ref: torch.Tensor, render: torch.Tensor, dists_model: Optional[torch.nn.Module] = None,
                adists_model: Optional[torch.nn.Module] = None, batch_size: int = 32, group=None,
                policy: Optional[str] = None, keep_aspect_ratio: bool = False) -> Dict[str, float]:
    """Score one video given as two (N,3,H,W) float32 tensors on the GPU, or -- with `policy` -- as two
    decoded uint8 (N,H,W,3) frame stacks on the GPU that are prepared per batch on the device
    (prep.prepare_frames: "interp256" = prep.py:89-95, "pil256" = prepare_image, ...).

    Frames are taken in batches of `batch_size`; with torch.distributed initialised, frame
    ranges shard across ranks and the scores are all-gathered once (sharding.py).  Argument
    order follows prep.py:186-189: model(ref, render); A-DISTS takes x = ref.
    """
    if ref.shape != render.shape:
        raise ValueError("ref and render differ in shape")
    n = ref.shape[0]
    out: Dict[str, float] = {}

    def run(model_call: Callable[[torch.Tensor, torch.Tensor], torch.Tensor]) -> np.ndarray:
        def batch(lo, hi):
            a, b = ref[lo:hi], render[lo:hi]
            if policy is not None:
                a = prep.prepare_frames(a, policy, keep_aspect_ratio=keep_aspect_ratio)
                b = prep.prepare_frames(b, policy, keep_aspect_ratio=keep_aspect_ratio)
            return model_call(a, b)

        scores = sharding.score_frames_sharded(batch, n, batch_size, ref.device, group)
        return scores.cpu().numpy()

    if adists_model is not None:
        out.update(video_columns("A-DISTS", run(lambda a, b: adists_model(a, b, as_loss=False))))
    if dists_model is not None:
        out.update(video_columns("DISTS", run(lambda a, b: dists_model(a, b, batch_average=False))))
    return out
