"""Authoring-container stand-in for the absent `wandb` package (oracle/make_goldens.py only).

The reference's training-time variants read a handful of hyper-parameters as attributes of
`wandb.config` (DISTS_pt_original.py:69-70,89-91,111-118; DISTS_pt_softmax.py:122;
model_stats.py:31-97).  This module is just that attribute bag; make_goldens sets the values.
"""


class _Config:
    pass


config = _Config()
