"""Drop-in `ADISTS` for nerf_qa.ADISTS.ADISTS, computed by libnqa_hip.so.

Mirrors nerf_qa/ADISTS/ADISTS.py:
  ADISTS(window_size=21)                                  :36-69
  .forward(x, y, as_loss=True, as_map=False)              :137-197
  .forward_once(x)                                        :112-125
Asymmetric like the reference: texture probabilities and entropy weights come from x
only (:147,153), so callers pass x = reference frame (prep.py:186).

Precision: the default here is "auto" = "f32s" for frames of at least 128x128 pixels -- float32 activations,
convolutions on the f16 matrix cores with both operands split into (hi, lo) half pairs (3 MFMAs per
product), 1e-7 from the exact-f32 path and from the reference at ~1.9x the exact-f32 ("f32") throughput --
and exact "f32" below (see AUTO_EXACT_PIXELS).  Plain 16-bit features are
NOT safe for A-DISTS: F.normalize (:166-167) and the entropy weights (:127-135) rescale every channel by
its own L2 norm, so a channel that is dead except for one pixel at 1e-4 is a full-scale feature
after normalisation, and the same channel rounded to exactly dead contributes T = S = 1 -- one such
flip moves the score by a channel weight, ~5e-4 (measured; DISTS' statistics have no such edge).
precision="f16" stays available as the opt-in fast mode (3x the throughput at 1080p).

as_loss=True in the reference runs the pyramids WITH autograd (:139-141).  Here, when a gradient is actually needed (grad mode
on and x or y requires grad), the tapped maps come from autograd.PyramidTaps (HIP forward, HIP backward through the 13 conv
layers in f32s, as for DISTS(require_grad=True)) and the head -- texture probabilities, entropy weights, windowed T / S -- is
evaluated in torch operations on the GPU (ADISTS/head.py) so that autograd differentiates it; the value returned is still the
fused kernel's.  Otherwise the value 1-mean(D) comes from the fused kernel alone (there is no graph to lose).  No script of
the reference calls it with a gradient (prep.py:186, test2_prep.py:151 pass as_loss=False).  as_map=True returns the
reference's [B,B,H,W] distortion map (:163,188-193) from one extra kernel.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import prec_id
from ..DISTS_pytorch.DISTS_pt import L2pooling as Downsample, _build_stages  # noqa: F401

from ..vgg_weights import load_vgg16_convs

# "auto" (the default): frames of fewer than AUTO_EXACT_PIXELS pixels run in "f32" (exact-f32 MFMA products, the
# reference's own 24-bit arithmetic), larger ones in "f32s" (22-23 bits, ~1.9x the conv throughput).  A-DISTS is
# discontinuous where a channel is dead except for a pixel or two (F.normalize, ADISTS.py:166-167): on frames of a few
# dozen pixels a side such channels are common and whether one survives can hinge on the last bit of a
# pre-activation -- round 2 found a 22x68 pair whose f32s score moved by 4.9e-4 when nothing but the summation order
# inside conv1_1 changed.  Below the threshold the cost of exact products is irrelevant (a 128x128 frame is
# launch-bound either way); above it no pair of ~5 900 random ones has moved by more than 6.4e-5
# (profiles/r02_stress_final.txt), and tools/gpu_stress.py now re-checks that with both conv1_1 forms.
DEFAULT_PRECISION = "auto"
AUTO_EXACT_PIXELS = 128 * 128
TWO_STREAM_MIN_PAIRS = 4          # batches of this many pairs or more ...
TWO_STREAM_MIN_PIXELS = 512 * 512  # ... of frames this large run as two half-batches on two HIP streams (_score)


class ADISTS(torch.nn.Module):
    def __init__(self, window_size=21, precision=None, vgg16_path=None):
        super().__init__()
        if window_size != 21:
            raise NotImplementedError("the HIP windowed-statistics kernel is built for the 21x21 window "
                                      "the reference always uses")
        convs, self.vgg_source = load_vgg16_convs(vgg16_path)
        self.stage1, self.stage2, self.stage3, self.stage4, self.stage5 = _build_stages(convs)
        for param in self.parameters():
            param.requires_grad = False
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, -1, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, -1, 1, 1))
        self.chns = [3, 64, 128, 256, 512, 512]
        self.windows = nn.ParameterList()
        self.window_size = window_size
        for k in range(len(self.chns)):
            self.windows.append(self.create_window(self.window_size, self.window_size / 3, self.chns[k]))
        self.precision = precision or os.environ.get("NQA_ADISTS_PRECISION", DEFAULT_PRECISION)
        if self.precision != "auto" and prec_id(self.precision) in (4, 5, 6, 7):
            raise ValueError("the mixed precisions ('f32m', 'f32m2', 'f32m4', 'f16w') are DISTS modes: A-DISTS needs float precision in every layer "
                             "(see the module docstring); use 'auto' (default), 'f32s', 'f32' or the opt-in 'f16'")
        self._packed = {}
        self._ws = ops.Workspace()
        self._side = {}

    # the window parameters are kept for state_dict compatibility (ADISTS.py:66-69,102-110);
    # the kernel uses the same separable 1-D Gaussian
    def gaussian(self, window_size, sigma):
        gauss = torch.Tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
                              for x in range(window_size)])
        return gauss / gauss.sum()

    def create_window(self, window_size, window_sigma, channel):
        _1d = self.gaussian(window_size, window_sigma).unsqueeze(1)
        _2d = _1d.mm(_1d.t()).float().unsqueeze(0).unsqueeze(0)
        return nn.Parameter(_2d.expand(channel, 1, window_size, window_size).contiguous(), requires_grad=False)

    def _conv_modules(self):
        return [m for st in (self.stage1, self.stage2, self.stage3, self.stage4, self.stage5)
                for m in st if isinstance(m, nn.Conv2d)]

    def precision_for(self, h: int, w: int) -> str:
        """The precision mode a frame size runs in ("auto": exact f32 below AUTO_EXACT_PIXELS pixels, f32s above)."""
        if self.precision != "auto":
            return self.precision
        return "f32" if h * w < AUTO_EXACT_PIXELS else "f32s"

    def _weights_key(self, dev):
        return (str(dev),) + tuple((m.weight._version, m.weight.data_ptr()) for m in self._conv_modules())

    def _packed_weights(self, dev, prec):
        convs = self._conv_modules()
        key = self._weights_key(dev)
        hit = self._packed.get(prec)
        if hit is None or hit[0] != key:
            blob = ops.pack_vgg_weights([(m.weight, m.bias) for m in convs], prec)
            hit = self._packed[prec] = (key, blob.to(dev))
        return hit[1]

    def __getstate__(self):
        d = self.__dict__.copy()
        for k in ("_packed", "_ws", "_side"):  # device scratch and streams never travel (__setstate__ rebuilds them)
            d.pop(k, None)
        return d

    def __setstate__(self, state):  # also a module pickled by the reference's class (install_alias())
        super().__setstate__(state)
        d = self.__dict__
        d.setdefault("precision", os.environ.get("NQA_ADISTS_PRECISION", DEFAULT_PRECISION))
        d.setdefault("vgg_source", "unpickled module (Conv2d weights of stage1..5)")
        d.pop("_packed_key", None)
        d["_packed"], d["_ws"], d["_side"] = {}, ops.Workspace(), {}

    def _score(self, x, y, prec):
        """D (B,) of a batch.  Large frames in batches of TWO_STREAM_MIN_PAIRS or more run as two half-batches on two
        HIP streams: the VALU-bound window pass of one half rides beside the MFMA-bound conv stack of the other (+2-3 %
        at 1080p, profiles/r03_adists_two_streams.txt; pairs are independent, so the scores are those of one call).
        NQA_ADISTS_STREAMS=1 switches it off."""
        dev, b = x.device, x.shape[0]
        packed = self._packed_weights(dev, prec)
        two = (dev.type == "cuda" and b >= TWO_STREAM_MIN_PAIRS and x.shape[-2] * x.shape[-1] >= TWO_STREAM_MIN_PIXELS
               and os.environ.get("NQA_ADISTS_STREAMS", "2") != "1")
        if not two:
            return ops.adists_forward(x, y, packed, prec, self._ws)
        side = self._side.get(str(dev))
        if side is None:
            side = self._side[str(dev)] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        main = torch.cuda.current_stream(dev)
        x, y = x.contiguous(), y.contiguous()  # (so that the halves are views)
        half, outs = (b + 1) // 2, []
        for st, (lo, hi) in zip(side, ((0, half), (half, b))):
            st.wait_stream(main)  # the frames (and the packed weights) were produced on the caller's stream
            with torch.cuda.stream(st):
                outs.append(ops.adists_forward(x[lo:hi], y[lo:hi], packed, prec, self._ws))  # (scratch is per stream)
        for st in side:
            main.wait_stream(st)
        return torch.cat(outs)

    def _loss_with_grad(self, x, y):
        """1 - mean(D) WITH its gradient towards x and y (ADISTS.py:139-141, 195).  The tapped maps come from
        autograd.PyramidTaps (HIP forward; HIP backward through the conv stack in f32s), the head is head.adists_d in
        torch operations, which autograd differentiates.  The VALUE returned is the scoring path's own (the fused
        kernel, same precision policy as as_loss=False), with the torch expression's gradient attached."""
        from .. import autograd
        from . import head
        tx = autograd.PyramidTaps.apply(x, self) if x.requires_grad else self._taps_nograd(x)
        ty = autograd.PyramidTaps.apply(y, self) if y.requires_grad else self._taps_nograd(y)
        d = head.adists_d([x.float()] + list(tx), [y.float()] + list(ty), self.window_size)
        loss = 1 - d.mean()
        with torch.no_grad():
            value = 1 - self._score(x.detach(), y.detach(), self.precision_for(x.shape[-2], x.shape[-1])).mean()
        return value + (loss - loss.detach())

    @torch.no_grad()
    def _taps_nograd(self, img):
        taps = ops.vgg_pyramid(img.float(), self._packed_weights(img.device, "f32s"), "f32s", self._ws)
        return [ops.nhwc_to_nchw_f32(t, "f32s") for t in taps]

    def forward_once(self, x):
        prec = self.precision_for(x.shape[-2], x.shape[-1])
        taps = ops.vgg_pyramid(x, self._packed_weights(x.device, prec), prec, self._ws)
        return [x] + [ops.nhwc_to_nchw_f32(t, prec) for t in taps]

    def forward(self, x, y, as_loss=True, as_map=False):
        assert x.shape == y.shape
        if as_loss and not as_map and torch.is_grad_enabled() and (x.requires_grad or y.requires_grad):
            return self._loss_with_grad(x, y)
        prec = self.precision_for(x.shape[-2], x.shape[-1])
        if as_map:
            # (:163,188-189,193) the reference's (B,H,W) + (B,1,H,W) addition broadcasts to (B,B,H,W)
            # with out[i, j] = map[i]; reproduced as is (callers use B = 1, nerf_nr_qa_prep_4.py:70)
            _, m = ops.adists_forward(x, y, self._packed_weights(x.device, prec), prec, self._ws, with_map=True)
            b = m.shape[0]
            return m.unsqueeze(1).expand(b, b, *m.shape[1:]).contiguous()
        d = self._score(x, y, prec)
        if as_loss:
            return 1 - d.mean()
        return 1 - d


def prepare_image(image, resize=True):
    """PIL image -> float32 (1,3,H,W) in [0,1]; ADISTS.py:200-204 (short side to 256, aspect ratio kept)."""
    from ..DISTS_pytorch.DISTS_pt import prepare_image as _p
    return _p(image, resize=resize, keep_aspect_ratio=True)
