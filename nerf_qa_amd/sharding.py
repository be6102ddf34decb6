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
