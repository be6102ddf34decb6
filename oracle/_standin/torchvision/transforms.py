"""The two transforms the reference's prepare_image uses (DISTS_pt.py:210-217, DISTS_pt_original.py:140-144),
backed by Pillow the way torchvision 0.17 documents them for PIL inputs -- authoring-container test tooling:

  functional.resize(img, size)  size=(h, w): img.resize((w, h), BILINEAR).  size=int: the SHORT side becomes
                                `size`, the long side int(size * long / short); unchanged if already that size.
  ToTensor()(img)               uint8 HWC -> float32 CHW / 255.

Only make_goldens.py's prepare_image goldens go through this; they pin WHICH resize the reference asks for
(policy and argument order), not Pillow's arithmetic, which tests/test_prep_oracle.py pins against Pillow itself.
"""
import numpy as np
import torch
from PIL import Image


class _Functional:
    @staticmethod
    def resize(img, size):
        w, h = img.size
        if isinstance(size, int):
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = size
        if (nw, nh) == (w, h):
            return img
        return img.resize((nw, nh), Image.BILINEAR)


functional = _Functional()


class ToTensor:
    def __call__(self, pic):
        arr = np.array(pic.convert("RGB") if pic.mode != "RGB" else pic, dtype=np.uint8, copy=True)
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
