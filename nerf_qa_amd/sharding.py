"""Frame sharding across the GPUs of one node.

Every frame pair is independent (no cross-frame state in DISTS_pt.py:105-148; the
per-video aggregation is a host-side mean, prep.py:191-198), so rank r of R scores the
contiguous range [r*ceil(N/R), (r+1)*ceil(N/R)) and the only exchange is ONE all-gather
of the per-rank float32 score vectors at the end of the video (RCCL over xGMI when the
backend is "nccl"; 10 k frames = 40 KB in total, so the collective is latency-bound).
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's frames; the last ranks may get fewer (or none)."""
    per = -(-n_frames // world)
    lo = min(n_frames, rank * per)
    return lo, min(n_frames, lo + per)


def gather_scores(local: torch.Tensor, n_frames: int, group=None) -> torch.Tensor:
    """All-gather the per-rank score vectors into the full (n_frames,) vector on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        assert local.numel() == n_frames
        return local
    world = dist.get_world_size(group)
    per = -(-n_frames // world)
    pad = torch.zeros(per, dtype=torch.float32, device=local.device)
    pad[: local.numel()] = local.float()
    out = torch.empty(world * per, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_frames]


def score_frames_sharded(score_batch: Callable[[int, int], torch.Tensor], n_frames: int, batch: int,
                         device, group=None) -> torch.Tensor:
    """Score frames [0, n_frames) across the process group.

    score_batch(lo, hi) returns the (hi-lo,) scores of frames lo..hi-1 on `device`
    (it loads / generates those frames itself).  Returns the full score vector on every rank.
    """
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(n_frames, rank, world)
    parts = []
    for s in range(lo, hi, batch):
        parts.append(score_batch(s, min(hi, s + batch)).reshape(-1).float())
    local = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float32, device=device)
    return gather_scores(local, n_frames, group)


def agree_precision(model, h: int, w: int, device, group=None) -> str:
    """DISTS `auto` under a process group: ONE verdict per job.  Rank 0 resolves the calibration report of the frame's
    size class (memory, the calibration file, or -- once per weight set, device model and library build -- a measurement
    on its GPU) and broadcasts it; the other ranks adopt it without launching anything, so an 8-rank job pays for one
    calibration, not eight, and the frames of one video are scored in one mode whatever the world size or which rank an
    empty shard lands on.  (Until round 3 every rank measured and the most accurate choice won; the measurement is
    deterministic for one device model and build, so rank 0's is as good as any.)  The agreed mode is kept on the module
    apart from the calibration cache (`model._agreed`), keyed by the weights it was agreed for.  Returns the mode frames
    of h x w run in; a module with a named precision, or no process group, is returned as it is."""
    if getattr(model, "precision", None) != "auto" or not (dist.is_available() and dist.is_initialized()):
        return model.precision_for(h, w, device)
    from .DISTS_pytorch.DISTS_pt import AUTO_MIN_PIXELS, size_class
    if h * w < AUTO_MIN_PIXELS:
        return model.precision_for(h, w, device)  # always f32s, nothing to agree on
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        choice = model.precision_for(h, w, device)  # (the live-alpha/beta verdict where that applies)
        box[0] = (choice, dict(model.calibrate(device, h, w)))
    # (an object broadcast: a few hundred bytes of pickled report; gloo and RCCL alike)
    on_gpu = torch.device(device).type == "cuda" and dist.get_backend(group) == "nccl"  # (RCCL moves device buffers only)
    dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group,
                               device=torch.device(device) if on_gpu else None)
    agreed, report = box[0]
    report = dict(report)
    report["choice_rank0_calibration"], report["choice"] = report.get("choice"), agreed
    report["agreed_over_ranks"], report["source"] = world, report.get("source", "?") + (" (rank 0)" if rank else "")
    if hasattr(model, "_agreed"):
        cls = max(size_class(h, w), 0)
        model._agreed[cls] = (model._weights_key(torch.device(device)), agreed)
        model._agreed_report = report
    return agreed
