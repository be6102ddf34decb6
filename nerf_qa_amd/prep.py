"""Frame preparation on the device: the reference's loader policies between a decoded uint8 frame
and the metric's float32 NCHW input, computed by libnqa_hip.so.

Mirrors, policy by policy:
  "interp256"      prep.py:89-95 load_video_frames: ToTensor (with the ToPILImage round trip of
                   :90-91) then F.interpolate(size=(256,256), bilinear, align_corners=False).
                   keep_aspect_ratio=True passes size=(256), which F.interpolate reads as the int
                   256 for BOTH dimensions, so the result is the same 256x256 -- reproduced as is.
  "interp"         data.py:80-82 / prep.py:132-133: /255 then F.interpolate to a given size.
  "equal_pixels"   test2_prep.py:424-437: ~65536 pixels at the source aspect ratio.  The reference
                   hands F.interpolate a float there, which its pinned torch (2.2.0) rejects with a
                   TypeError; int() of that float is used here (documented deviation).
  "pil256"         DISTS_pt.py:210-217 prepare_image(resize=True): PIL resize to 256x256 when the
                   short side exceeds 256; keep_aspect_ratio=True -> short side 256
                   (torchvision's rule: long side = int(256 * long / short)).
  "full"           ToTensor only (test2_prep.py:324-326, prepare_image(resize=False)).
Inputs are uint8 (n,H,W,3) device tensors (a decoded RGB frame batch); outputs float32 (n,3,h,w).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import ops

POLICIES = ("interp256", "interp", "equal_pixels", "pil256", "full")


def equal_pixel_size(oh: int, ow: int) -> Tuple[int, int]:
    """test2_prep.py:428-435 (with the int() the pinned torch would have required)."""
    if ow >= oh:
        ratio = float(ow) / float(oh)
        h = math.sqrt(256 * 256 / ratio)
        w = int(ratio * h)
        return int(h), w
    ratio = float(oh) / float(ow)
    w = math.sqrt(256 * 256 / ratio)
    h = int(ratio * w)
    return h, int(w)


def pil_resize_size(oh: int, ow: int, keep_aspect_ratio: bool) -> Tuple[int, int]:
    """Output (H, W) of prepare_image's resize (DISTS_pt.py:211-215); unchanged when min side <= 256."""
    if min(oh, ow) <= 256:
        return oh, ow
    if not keep_aspect_ratio:
        return 256, 256
    short, long = (ow, oh) if ow <= oh else (oh, ow)
    new_long = int(256 * long / short)  # torchvision.transforms.functional._compute_resized_output_size
    return (new_long, 256) if ow <= oh else (256, new_long)


def prepare_frames(frames_u8: torch.Tensor, policy: str = "interp256", size=None, keep_aspect_ratio: bool = False,
                   ws: ops.Workspace | None = None) -> torch.Tensor:
    """uint8 (n,H,W,3) on the GPU -> float32 (n,3,h,w) per the named reference policy."""
    n, oh, ow, _ = frames_u8.shape
    if policy == "interp256":
        # (the ToPILImage -> ToTensor round trip of :90-91 is the identity in float32, see
        # tests/test_prep_oracle.py, so the fused ToTensor+interpolate kernel is exact here)
        return ops.u8_resize_bilinear_f32(frames_u8, (256, 256))
    if policy == "interp":
        if size is None:
            raise ValueError("policy 'interp' needs size=(H, W) or an int")
        return ops.u8_resize_bilinear_f32(frames_u8, size)
    if policy == "equal_pixels":
        return ops.u8_resize_bilinear_f32(frames_u8, equal_pixel_size(oh, ow))
    if policy == "pil256":
        h, w = pil_resize_size(oh, ow, keep_aspect_ratio)
        if (h, w) != (oh, ow):
            frames_u8 = ops.resize_pil_bilinear_u8(frames_u8, (h, w), ws)
        return ops.u8hwc_to_f32nchw(frames_u8)
    if policy == "full":
        return ops.u8hwc_to_f32nchw(frames_u8)
    raise ValueError(f"unknown policy {policy!r}; one of {POLICIES}")
