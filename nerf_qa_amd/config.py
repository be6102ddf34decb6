"""Run configuration the reference reads from the global `wandb.config`.

DISTS_pt_original / DISTS_pt_softmax / model_stats in the reference read options from
`wandb.config` inside model code (DISTS_pt_original.py:69-70,111-119; model_stats.py:31-66;
model.py:26,40,52).
If wandb is importable and a run is active, that object is used, so the reference's scripts keep
working unchanged; otherwise this module-level namespace holds the same keys with the defaults
of wandb/config-nerf-qa.yaml-style runs and can be edited by the caller.
"""
from __future__ import annotations

from types import SimpleNamespace

_local = SimpleNamespace(
    weight_lower_bound=0.0,
    alpha_beta_ratio=1.0,
    dists_weight_norm="off",      # '+'-joined flags: relu, w_sum_detach ; or 'softmax' (model_stats.py:57)
    detach_beta="False",
    regression_type="linear",     # linear | sqrt | logistic
    subjective_score_type="MOS",
    mode="linear",                # nerf_qa/model.py:26,40,52: linear | sqrt | softmax | softmax+sqrt
)


def config():
    try:
        import wandb  # type: ignore
        if getattr(wandb, "run", None) is not None:
            return wandb.config
    except Exception:
        pass
    return _local
