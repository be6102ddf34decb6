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
    """DISTS `auto` under a process group: every rank calibrates on its own GPU (the measurement is deterministic, but
    two GPUs may land on different sides of a threshold), then ALL ranks adopt the most accurate rung any rank chose, so
    the frames of one video are scored in one mode whatever the world size.  Returns the mode frames of h x w run in;
    a module with a named precision, or no process group, is returned as it is."""
    prec = model.precision_for(h, w, device)
    if getattr(model, "precision", None) != "auto" or not (dist.is_available() and dist.is_initialized()):
        return prec
    from .DISTS_pytorch.DISTS_pt import AUTO_MIN_PIXELS, LADDER
    if h * w < AUTO_MIN_PIXELS:
        return prec  # always f32s, nothing to agree on
    t = torch.tensor([LADDER.index(prec)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)  # LADDER is ordered fastest -> most accurate
    agreed = LADDER[int(t.item())]
    report = model.calibrate(device, h, w)  # (the cached report of this size class: what precision_for reads)
    if agreed != prec:
        report["choice_local"], report["choice"] = prec, agreed
    report["agreed_over_ranks"] = dist.get_world_size(group)
    return agreed
