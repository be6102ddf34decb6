"""Per-video scoring harness: the build's counterpart of the reference's offline loops
(prep.py:181-216, test2_prep.py:146-193,252-296,352-396,464-512): frames -> batched A-DISTS + DISTS
-> per-video columns -> CSV.  Host-side glue only; the scores come from the HIP modules.

Column arithmetic follows the reference to the type: the per-frame scores are the float32 arrays
the modules return (`.detach().cpu().numpy()`, test2_prep.py:152-155), so np.mean / np.std /
np.min / np.max are taken IN float32 and return np.float32 (:158-165), and
`frame_bias = video_score - frame_scores` is a float32 array (:167-168) serialised by `to_str`
(:123-125) as "['1.000000e-01', ...]".  `frame_count` is `len(frames_data)`, i.e. the number of
BATCHES of the DataLoader (:181), not of frames -- reproduced as written.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch

from . import prep, sharding

# column-name suffix per resize policy of test2_prep.py (:183-193, :286-296, :386-396, :502-509)
POLICY_SUFFIX = {None: "", "pil256_aspect": "", "full": "_full_size", "pil256": "_square",
                 "equal_pixels": "_pixel_count"}


def to_str(array) -> str:
    """test2_prep.py:123-125: the list of '{:.6e}' strings, as Python prints a list of str."""
    return str(["{:.6e}".format(num) for num in array])


def video_columns(name: str, frame_scores: np.ndarray, suffix: str = "") -> Dict[str, np.float32]:
    """The four per-video columns of one metric (prep.py:191-216, test2_prep.py:158-165,183-190):
    `name+suffix`, `..._std`, `..._min`, `..._max`, computed in the scores' own dtype (float32)."""
    s = np.asarray(frame_scores)
    if s.dtype != np.float32:
        s = s.astype(np.float32)
    base = name + suffix
    return {base: np.mean(s), f"{base}_std": np.std(s), f"{base}_min": np.min(s), f"{base}_max": np.max(s)}


def frame_bias(frame_scores: np.ndarray) -> np.ndarray:
    """test2_prep.py:167-168: video score (float32 mean) minus every frame's score."""
    s = np.asarray(frame_scores, dtype=np.float32)
    return np.mean(s) - s


def synthetic_frames(indices, h: int, w: int, device):
    """Frames `indices` of the synthetic video of BASELINE.json configs[3] / SURVEY.md 8d, generated ON THE DEVICE:
    frame i is a pure function of (i, h, w) -- the device generator is re-seeded with seed = frame index -- so any
    rank can produce any frame range without the video ever existing in host or device memory as a whole.
    Returns (ref, render) float32 (n,3,h,w): ref = U[0,1), render = clamp(ref + 0.1 N(0,1), 0, 1)."""
    idx = [int(i) for i in indices]
    ref = torch.empty(len(idx), 3, h, w, dtype=torch.float32, device=device)
    ren = torch.empty_like(ref)
    g = torch.Generator(device=device)
    for j, i in enumerate(idx):
        g.manual_seed(i)
        ref[j].uniform_(0.0, 1.0, generator=g)
        ren[j].normal_(0.0, 0.1, generator=g)
    ren.add_(ref).clamp_(0.0, 1.0)
    return ref, ren


@torch.no_grad()
def score_video(ref: torch.Tensor, render: torch.Tensor, dists_model: Optional[torch.nn.Module] = None,
                adists_model: Optional[torch.nn.Module] = None, batch_size: int = 8, group=None,
                policy: Optional[str] = None, keep_aspect_ratio: bool = False, suffix: str = "",
                with_frame_bias: bool = True, return_frame_scores: bool = False) -> Dict[str, object]:
    """Score one video given as two (N,3,H,W) float32 tensors on the GPU, or -- with `policy` -- as two
    decoded uint8 (N,H,W,3) frame stacks on the GPU that are prepared per batch on the device
    (prep.prepare_frames: "interp256" = prep.py:89-95, "pil256" = prepare_image, ...).

    Frames are taken in batches of `batch_size` (test2_prep.py:120 uses 8); with torch.distributed
    initialised, frame ranges shard across ranks and the scores are all-gathered once (sharding.py).
    Argument order follows prep.py:186-189: model(ref, render); A-DISTS takes x = ref.
    Returns the reference's columns for this video: `A-DISTS`/`DISTS` (+`suffix`) with _std/_min/_max,
    and when `with_frame_bias` (test2_prep.py's first pass, suffix "") `frame_count`,
    `frame_bias_adists`, `frame_bias_dists`.
    """
    if ref.shape != render.shape:
        raise ValueError("ref and render differ in shape")
    n = ref.shape[0]
    out: Dict[str, object] = {}

    def run(model_call: Callable[[torch.Tensor, torch.Tensor], torch.Tensor]) -> np.ndarray:
        def batch(lo, hi):
            a, b = ref[lo:hi], render[lo:hi]
            if policy is not None:
                a = prep.prepare_frames(a, policy, keep_aspect_ratio=keep_aspect_ratio)
                b = prep.prepare_frames(b, policy, keep_aspect_ratio=keep_aspect_ratio)
            return model_call(a, b)

        scores = sharding.score_frames_sharded(batch, n, batch_size, ref.device, group)
        return scores.cpu().numpy()

    frames = {}
    if dists_model is not None and getattr(dists_model, "precision", None) == "auto" and policy is None:
        # one precision mode per video: under a process group rank 0's calibration verdict is broadcast BEFORE the
        # shards are scored (ranks whose shard is empty never enter forward(), so the collective cannot live there);
        # with no process group this is the plain precision_for.  (With `policy` the prepared size is only known per
        # batch; callers who shard prepared-on-the-fly videos call sharding.agree_precision with that size themselves.)
        sharding.agree_precision(dists_model, int(ref.shape[-2]), int(ref.shape[-1]), ref.device, group)
    if adists_model is not None:
        frames["A-DISTS"] = run(lambda a, b: adists_model(a, b, as_loss=False))
        out.update(video_columns("A-DISTS", frames["A-DISTS"], suffix))
    if dists_model is not None:
        frames["DISTS"] = run(lambda a, b: dists_model(a, b, batch_average=False))
        out.update(video_columns("DISTS", frames["DISTS"], suffix))
    if with_frame_bias:
        out["frame_count"] = -(-n // batch_size)  # len(DataLoader): batches, as the reference counts them
        if "A-DISTS" in frames:
            out["frame_bias_adists"] = to_str(frame_bias(frames["A-DISTS"]))
        if "DISTS" in frames:
            out["frame_bias_dists"] = to_str(frame_bias(frames["DISTS"]))
    if return_frame_scores:
        out["_frame_scores"] = frames
    return out


# the order test2_prep.py:183-193 assigns the first pass's columns in
COLUMN_ORDER = ("A-DISTS", "DISTS", "A-DISTS_std", "DISTS_std", "A-DISTS_min", "DISTS_min", "A-DISTS_max",
                "DISTS_max", "frame_count", "frame_bias_adists", "frame_bias_dists")


def add_video_columns(test_df, per_video: Iterable[Dict[str, object]], suffix: str = ""):
    """Assign the per-video results to the score table the way test2_prep.py:183-193 (and :286-296 etc. with a
    suffix) does: one new column per key, in the reference's order, each a list over the table's rows."""
    rows = list(per_video)
    if len(rows) != len(test_df):
        raise ValueError(f"{len(rows)} video results for a table of {len(test_df)} rows")
    order = [c if c.startswith("frame_") else c.replace("DISTS", "DISTS" + suffix, 1) for c in COLUMN_ORDER]
    for col in order:
        if rows and col in rows[0]:
            test_df[col] = [r[col] for r in rows]
    return test_df


def write_scores_csv(test_df, path: str) -> None:
    """test2_prep.py:512 / data_prep.py:120: `test_df.to_csv(path)` (index column included)."""
    test_df.to_csv(path)
