"""Tensor-level wrappers over the C ABI (include/nqa.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
number is produced by the kernels in libnqa_hip.so.  All functions require CUDA (ROCm)
tensors and raise otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import NUM_CONVS, PREC_DTYPE, TOTAL_CHNS, check, lib, prec_id, ptr, stream_ptr

CHNS = (3, 64, 128, 256, 512, 512)
CONV_COUT = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
CONV_CIN = (3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512)
CONV_STAGE = (0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4)


def _need_cuda(*ts: torch.Tensor) -> torch.device:
    dev = ts[0].device
    for t in ts:
        if not t.is_cuda:
            raise _lib.NqaError("nerf_qa_amd runs on the GPU only: got a tensor on %s "
                                "(move inputs and the module to cuda; there is no CPU fallback)" % t.device)
        if t.device != dev:
            raise _lib.NqaError("tensors on different devices")
    return dev


def _on(dev: torch.device):
    """The library launches on the current HIP device: make that the tensors' device for the duration of the
    call and restore the caller's afterwards (a process that drives several GPUs from one thread would
    otherwise launch on the wrong one -- or, if we left it switched, allocate on the wrong one later)."""
    return torch.cuda.device(dev)


def _call(dev: torch.device, fn, *args) -> None:
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with _on(dev):
            check(fn(*args))
    else:
        check(fn(*args))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def pyramid_dims(h: int, w: int):
    dims = [(h, w)]
    for _ in range(4):
        h, w = (h + 1) // 2, (w + 1) // 2
        dims.append((h, w))
    return dims


def pack_vgg_weights(convs: Sequence, prec) -> torch.Tensor:
    """13 (weight OIHW, bias) pairs (numpy or torch, any device) -> packed blob (CPU uint8 tensor)."""
    p = prec_id(prec)
    if len(convs) != NUM_CONVS:
        raise ValueError("expected 13 conv layers")
    ws, bs = [], []
    for li, (w, b) in enumerate(convs):
        w = np.ascontiguousarray(w.detach().cpu().numpy() if torch.is_tensor(w) else w, dtype=np.float32)
        b = np.ascontiguousarray(b.detach().cpu().numpy() if torch.is_tensor(b) else b, dtype=np.float32)
        if w.shape != (CONV_COUT[li], CONV_CIN[li], 3, 3) or b.shape != (CONV_COUT[li],):
            raise ValueError(f"conv {li}: unexpected shape {w.shape} / {b.shape}")
        ws.append(w)
        bs.append(b)
    nbytes = lib().nqa_packed_weights_bytes(p)
    blob = torch.empty(nbytes, dtype=torch.uint8)
    wp = (C.c_void_p * NUM_CONVS)(*[w.ctypes.data for w in ws])
    bp = (C.c_void_p * NUM_CONVS)(*[b.ctypes.data for b in bs])
    check(lib().nqa_pack_vgg_weights(wp, bp, p, blob.data_ptr()))
    return blob


TAP_LAYERS = (1, 3, 6, 9, 12)  # conv layers whose output is tapped (relu1_2 .. relu5_3)


def conv1_1(x: torch.Tensor, packed: torch.Tensor, prec) -> torch.Tensor:
    """relu1_1 as NHWC in prec's activation format ("f32s": split16 bytes in a float32 tensor, see
    include/nqa.h Layouts; split16_decode turns them into floats)."""
    p = prec_id(prec)
    dev = _need_cuda(x, packed)
    x = _f32c(x)
    n, c, h, w = x.shape
    assert c == 3
    out = torch.empty((n, h, w, 64), dtype=PREC_DTYPE[p], device=dev)
    _call(dev, lib().nqa_conv1_1, ptr(x), n, h, w, ptr(packed), p, ptr(out), stream_ptr(dev))
    return out


def conv1_fused(x: torch.Tensor, packed: torch.Tensor, prec) -> torch.Tensor:
    """relu1_2 (NHWC) straight from the image: conv1_1 + conv1_2 in one kernel (16-bit modes and "f32s": float out)."""
    p = prec_id(prec)
    dev = _need_cuda(x, packed)
    x = _f32c(x)
    n, c, h, w = x.shape
    assert c == 3
    out = torch.empty((n, h, w, 64), dtype=PREC_DTYPE[p], device=dev)
    _call(dev, lib().nqa_conv1_fused, ptr(x), n, h, w, ptr(packed), p, ptr(out), stream_ptr(dev))
    return out


def conv3x3_relu(inp: torch.Tensor, layer: int, packed: torch.Tensor, prec) -> torch.Tensor:
    """One VGG conv layer, NHWC.  "f32s": split16 in; float out for TAP_LAYERS, split16 out otherwise."""
    p = prec_id(prec)
    dev = _need_cuda(inp, packed)
    assert inp.dtype == PREC_DTYPE[p] and inp.is_contiguous()
    n, h, w, c = inp.shape
    assert c == CONV_CIN[layer]
    out = torch.empty((n, h, w, CONV_COUT[layer]), dtype=inp.dtype, device=dev)
    _call(dev, lib().nqa_conv3x3_relu, ptr(inp), n, h, w, layer, ptr(packed), p, ptr(out), stream_ptr(dev))
    return out


def l2pool(inp: torch.Tensor, prec) -> torch.Tensor:
    """L2-pool of a tapped map, NHWC.  "f32s": float in, split16 out (it feeds the next conv)."""
    p = prec_id(prec)
    dev = _need_cuda(inp)
    assert inp.dtype == PREC_DTYPE[p] and inp.is_contiguous()
    n, h, w, c = inp.shape
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=inp.dtype, device=dev)
    _call(dev, lib().nqa_l2pool, ptr(inp), n, h, w, c, p, ptr(out), stream_ptr(dev))
    return out


def conv_pool_stats(inp: torch.Tensor, layer: int, packed: torch.Tensor, prec):
    """The stage-closing conv `layer` + ReLU + L2-pool + statistics sums in ONE kernel (include/nqa.h,
    nqa_conv_pool_stats): `inp` is the 2B-image NHWC batch of the layer's input (x images then y images); returns
    (pooled (2B, ceil(H/2), ceil(W/2), Cout) in the next stage's input dtype, sums float64 (B, Cout, 5) = sum x, sum y,
    sum x^2, sum y^2, sum xy over the tap's pixels).  Raises NqaError where no fused form exists (only conv2_2 so far)."""
    p = prec_id(prec)
    dev = _need_cuda(inp, packed)
    assert inp.is_contiguous() and inp.shape[0] % 2 == 0
    n, h, w, c = inp.shape
    assert c == CONV_CIN[layer]
    b = n // 2
    cout = CONV_COUT[layer]
    pooled = torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout), dtype=PREC_DTYPE[_lib.stage_prec(p, CONV_STAGE[layer] + 1)]
                         if p in _lib.MIXED_STAGES else inp.dtype, device=dev)
    sums = torch.empty((b, cout, 5), dtype=torch.float64, device=dev)
    nbytes = lib().nqa_conv_pool_workspace_bytes(b, h, w, layer)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    _call(dev, lib().nqa_conv_pool_stats, ptr(inp), b, h, w, layer, ptr(packed), p, ptr(pooled), ptr(sums), ptr(ws), ws.numel(),
          stream_ptr(dev))
    return pooled, sums


def conv1_pool_stats(x: torch.Tensor, y: torch.Tensor, packed: torch.Tensor, prec):
    """Stage 1 + L2-pool + tap-1 statistics in ONE kernel from the raw frames (include/nqa.h, nqa_conv1_pool_stats):
    x, y (B,3,H,W) float32 -> (pooled relu1_2 (2B, ceil(H/2), ceil(W/2), 64) f16 NHWC, x images first; sums float64 (B, 64, 5))."""
    p = prec_id(prec)
    dev = _need_cuda(x, y, packed)
    x, y = _f32c(x), _f32c(y)
    b, c, h, w = x.shape
    assert c == 3 and y.shape == x.shape
    pooled = torch.empty((2 * b, (h + 1) // 2, (w + 1) // 2, 64), dtype=torch.float16, device=dev)
    sums = torch.empty((b, 64, 5), dtype=torch.float64, device=dev)
    nbytes = lib().nqa_conv_pool_workspace_bytes(b, h, w, 1)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    _call(dev, lib().nqa_conv1_pool_stats, ptr(x), ptr(y), b, h, w, ptr(packed), p, ptr(pooled), ptr(sums), ptr(ws), ws.numel(),
          stream_ptr(dev))
    return pooled, sums


def nhwc_to_nchw_f32(inp: torch.Tensor, prec) -> torch.Tensor:
    p = prec_id(prec)
    dev = _need_cuda(inp)
    assert inp.dtype == PREC_DTYPE[p] and inp.is_contiguous()
    n, h, w, c = inp.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_nhwc_to_nchw_f32, ptr(inp), n, h, w, c, p, ptr(out), stream_ptr(dev))
    return out


class Workspace:
    """Grow-only device scratch, one buffer per (module, device, HIP stream): avoids allocator traffic per call, and two
    streams that run the same module side by side (ADISTS' two half-batches) never share scratch.  At most MAX_STREAMS
    buffers are kept (the least recently used one goes first): a caller that scores from many short-lived streams does
    not pile up gigabytes of scratch."""

    MAX_STREAMS = 8

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, dev: torch.device) -> torch.Tensor:
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        buf = self.bufs.pop(key, None)
        if buf is None or buf.numel() < nbytes:
            buf = None  # (released before the larger one is requested)
            while len(self.bufs) >= self.MAX_STREAMS:
                self.bufs.pop(next(iter(self.bufs)))
            buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        self.bufs[key] = buf  # (re-inserted: most recently used last)
        return buf


def tap_prec(prec, k: int) -> int:
    """Precision id of tapped map k (0 = relu1_2 .. 4 = relu5_3) in mode `prec` ("f32m": half, half, half, float, float)."""
    return _lib.stage_prec(prec_id(prec), k)


def vgg_pyramid(x: torch.Tensor, packed: torch.Tensor, prec, ws: Workspace | None = None) -> List[torch.Tensor]:
    """Five tapped maps relu1_2..relu5_3 as NHWC tensors in prec's dtype (per tap in "f32m", see tap_prec)."""
    p = prec_id(prec)
    dev = _need_cuda(x, packed)
    x = _f32c(x)
    n, c, h, w = x.shape
    assert c == 3
    taps = [torch.empty((n, hk, wk, ck), dtype=PREC_DTYPE[_lib.stage_prec(p, k)], device=dev)
            for k, ((hk, wk), ck) in enumerate(zip(pyramid_dims(h, w), CHNS[1:]))]
    nbytes = lib().nqa_workspace_bytes(n, h, w, p)
    buf = (ws or Workspace()).get(nbytes, dev)
    tp = (C.c_void_p * 5)(*[ptr(t) for t in taps])
    _call(dev, lib().nqa_vgg_pyramid, ptr(x), n, h, w, ptr(packed), p, ptr(buf), buf.numel(), tp, stream_ptr(dev))
    return taps


def _max_pairs(bytes_for, b: int) -> int:
    """Slice size for a batch whose workspace would exceed NQA_MAX_WORKSPACE_GB (default 96, a third of an
    MI355X's HBM): the batch then runs in equal slices (pairs are independent, so the results are the same)."""
    budget = int(float(os.environ.get("NQA_MAX_WORKSPACE_GB", "96")) * (1 << 30))
    if b <= 1 or bytes_for(b) <= budget:
        return b
    lo, hi = 1, b  # bytes_for is monotone in the batch size
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if bytes_for(mid) <= budget:
            lo = mid
        else:
            hi = mid - 1
    slices = -(-b // lo)
    return -(-b // slices)  # equal slices rather than full ones plus a remainder


def dists_forward(x: torch.Tensor, y: torch.Tensor, packed: torch.Tensor, prec, ws: Workspace | None = None):
    """(S1, S2), each float32 (B, 1475): both pyramids + statistics in one enqueue."""
    p = prec_id(prec)
    dev = _need_cuda(x, y, packed)
    x, y = _f32c(x), _f32c(y)
    if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected two (B,3,H,W) tensors of equal shape, got {tuple(x.shape)} / {tuple(y.shape)}")
    b, _, h, w = x.shape
    mb = _max_pairs(lambda n: lib().nqa_workspace_bytes(2 * n, h, w, p), b)
    if mb < b:
        parts = [dists_forward(x[i:i + mb], y[i:i + mb], packed, prec, ws) for i in range(0, b, mb)]
        return torch.cat([q[0] for q in parts]), torch.cat([q[1] for q in parts])
    s1 = torch.empty((b, TOTAL_CHNS), dtype=torch.float32, device=dev)
    s2 = torch.empty((b, TOTAL_CHNS), dtype=torch.float32, device=dev)
    nbytes = lib().nqa_workspace_bytes(2 * b, h, w, p)
    buf = (ws or Workspace()).get(nbytes, dev)
    _call(dev, lib().nqa_dists_forward, ptr(x), ptr(y), b, h, w, ptr(packed), p, ptr(buf), buf.numel(), ptr(s1), ptr(s2),
                                  stream_ptr(dev))
    return s1, s2


def dists_stats_nchw(feats0: Sequence[torch.Tensor], feats1: Sequence[torch.Tensor]):
    """(S1, S2) from two lists of six float32 NCHW feature maps (forward_from_feats)."""
    if len(feats0) != 6 or len(feats1) != 6:
        raise ValueError("expected six feature maps per image")
    dev = _need_cuda(*feats0, *feats1)
    f0 = [_f32c(f) for f in feats0]
    f1 = [_f32c(f) for f in feats1]
    b = f0[0].shape[0]
    for a, c in zip(f0, f1):
        if a.shape != c.shape or a.dim() != 4 or a.shape[0] != b:
            raise ValueError("feature lists disagree in shape")
    cs = (C.c_int * 6)(*[f.shape[1] for f in f0])
    hs = (C.c_int * 6)(*[f.shape[2] for f in f0])
    wsz = (C.c_int * 6)(*[f.shape[3] for f in f0])
    ctot = sum(f.shape[1] for f in f0)
    s1 = torch.empty((b, ctot), dtype=torch.float32, device=dev)
    s2 = torch.empty((b, ctot), dtype=torch.float32, device=dev)
    nbytes = lib().nqa_stats_scratch_bytes(b, cs, hs, wsz)
    scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    p0 = (C.c_void_p * 6)(*[ptr(f) for f in f0])
    p1 = (C.c_void_p * 6)(*[ptr(f) for f in f1])
    _call(dev, lib().nqa_dists_stats_nchw, p0, p1, b, cs, hs, wsz, ptr(scratch), scratch.numel(), ptr(s1), ptr(s2),
                                     stream_ptr(dev))
    return s1, s2


def dists_score(s1: torch.Tensor, s2: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """score (B,) = 1 - sum(alpha*S1 + beta*S2)/(sum alpha + sum beta), fused kernel (no autograd)."""
    dev = _need_cuda(s1, s2, alpha, beta)
    s1, s2 = _f32c(s1), _f32c(s2)
    a, b_ = _f32c(alpha.detach().reshape(-1)), _f32c(beta.detach().reshape(-1))
    assert s1.shape == s2.shape and s1.shape[1] == TOTAL_CHNS == a.numel() == b_.numel()
    out = torch.empty((s1.shape[0],), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_dists_score, ptr(s1), ptr(s2), ptr(a), ptr(b_), s1.shape[0], ptr(out), stream_ptr(dev))
    return out


DEFAULT_CONV_VARIANT = 1  # the library's start-up value (include/nqa.h)


def set_conv_variant(v: int) -> None:
    check(lib().nqa_set_conv_variant(int(v)))


def dists_fused_taps(b: int, h: int, w: int, prec) -> tuple:
    """Taps (1..5) whose L2-pool and statistics nqa_dists_forward runs inside the stage-closing conv kernel for this
    shape and mode (the calling thread's conv variant bits 64 / 128 switch them off)."""
    f = (C.c_int * 6)()
    check(lib().nqa_dists_fused_taps(int(b), int(h), int(w), prec_id(prec), f))
    return tuple(k for k in range(6) if f[k])


def timing_enable(on: bool) -> None:
    check(lib().nqa_timing_enable(1 if on else 0))


def timing_collect():
    n = (C.c_int * len(_lib.K_NAMES))()
    ms = (C.c_double * len(_lib.K_NAMES))()
    check(lib().nqa_timing_collect(n, ms))
    return {name: (n[i], ms[i]) for i, name in enumerate(_lib.K_NAMES)}


def adists_forward(x: torch.Tensor, y: torch.Tensor, packed: torch.Tensor, prec, ws: Workspace | None = None,
                   with_map: bool = False):
    """D (B,) float32 of ADISTS.forward (ADISTS.py:147-191); the caller returns 1-D or 1-mean(D).

    with_map=True also returns the as_map=True distortion map (ADISTS.py:188-189,193) as (B,H,W)."""
    p = prec_id(prec)
    dev = _need_cuda(x, y, packed)
    x, y = _f32c(x), _f32c(y)
    if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected two (B,3,H,W) tensors of equal shape, got {tuple(x.shape)} / {tuple(y.shape)}")
    b, _, h, w = x.shape
    mb = _max_pairs(lambda n: lib().nqa_adists_workspace_bytes(n, h, w, p), b)
    if mb < b:
        parts = [adists_forward(x[i:i + mb], y[i:i + mb], packed, prec, ws, with_map) for i in range(0, b, mb)]
        if with_map:
            return torch.cat([q[0] for q in parts]), torch.cat([q[1] for q in parts])
        return torch.cat(parts)
    d = torch.empty((b,), dtype=torch.float32, device=dev)
    nbytes = lib().nqa_adists_workspace_bytes(b, h, w, p)
    buf = (ws or Workspace()).get(nbytes, dev)
    if with_map:
        m = torch.empty((b, h, w), dtype=torch.float32, device=dev)
        _call(dev, lib().nqa_adists_forward_map, ptr(x), ptr(y), b, h, w, ptr(packed), p, ptr(buf), buf.numel(), ptr(d),
                                           ptr(m), stream_ptr(dev))
        return d, m
    _call(dev, lib().nqa_adists_forward, ptr(x), ptr(y), b, h, w, ptr(packed), p, ptr(buf), buf.numel(), ptr(d),
                                   stream_ptr(dev))
    return d


# ---- input preparation ---------------------------------------------------
def u8hwc_to_f32nchw(frames: torch.Tensor, pil_roundtrip: bool = False) -> torch.Tensor:
    """ToTensor on the device: uint8 (n,H,W,3) -> float32 (n,3,H,W) / 255 (prep.py:89, data.py:80)."""
    dev = _need_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError(f"expected uint8 (n,H,W,3), got {frames.dtype} {tuple(frames.shape)}")
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_u8hwc_to_f32nchw, ptr(frames), n, h, w, int(pil_roundtrip), ptr(out), stream_ptr(dev))
    return out


def resize_bilinear_f32(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for float32 (n,C,H,W)."""
    dev = _need_cuda(x)
    x = _f32c(x)
    if x.dim() != 4:
        raise ValueError(f"expected (n,C,H,W), got {tuple(x.shape)}")
    ho, wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    n, c, h, w = x.shape
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_resize_bilinear_f32, ptr(x), n * c, h, w, ho, wo, ptr(out), stream_ptr(dev))
    return out


def u8_resize_bilinear_f32(frames: torch.Tensor, size) -> torch.Tensor:
    """ToTensor + F.interpolate(bilinear, align_corners=False) fused: uint8 (n,H,W,3) -> float32 (n,3,h,w)."""
    dev = _need_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError(f"expected uint8 (n,H,W,3), got {frames.dtype} {tuple(frames.shape)}")
    frames = frames.contiguous()
    ho, wo = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    n, h, w, _ = frames.shape
    out = torch.empty((n, 3, ho, wo), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_u8_resize_bilinear_f32, ptr(frames), n, h, w, ho, wo, ptr(out), stream_ptr(dev))
    return out


def resize_pil_bilinear_u8(frames: torch.Tensor, size, ws: Workspace | None = None) -> torch.Tensor:
    """PIL Image.resize((W,H), BILINEAR) on uint8 (n,H,W,3) frames, bit-exact; size = (Hout, Wout)."""
    dev = _need_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError(f"expected uint8 (n,H,W,3), got {frames.dtype} {tuple(frames.shape)}")
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((n, ho, wo, 3), dtype=torch.uint8, device=dev)
    nbytes = lib().nqa_resize_pil_workspace_bytes(n, h, w, ho, wo)
    buf = (ws or Workspace()).get(nbytes, dev)
    _call(dev, lib().nqa_resize_pil_bilinear_u8, ptr(frames), n, h, w, ho, wo, ptr(buf), buf.numel(), ptr(out),
                                           stream_ptr(dev))
    return out


# ---- split16 (NQA_PREC_F32S activation format between conv layers; tests and tools only) ----------
def split16_encode(a: torch.Tensor) -> torch.Tensor:
    """float32 NHWC (..., C) -> split16 bytes viewed as float32 of the same shape (4 bytes per element)."""
    dev = _need_cuda(a)
    a = _f32c(a)
    c = a.shape[-1]
    out = torch.empty_like(a)
    _call(dev, lib().nqa_split16_encode, ptr(a), a.numel() // c, c, ptr(out), stream_ptr(dev))
    return out


def split16_decode(a: torch.Tensor) -> torch.Tensor:
    dev = _need_cuda(a)
    assert a.dtype == torch.float32 and a.is_contiguous()
    c = a.shape[-1]
    out = torch.empty_like(a)
    _call(dev, lib().nqa_split16_decode, ptr(a), a.numel() // c, c, ptr(out), stream_ptr(dev))
    return out


# ---- backward pass of the DISTS pyramid (require_grad=True; include/nqa.h, nqa_backward.hip) -----------------------
def pack_conv_split(w) -> torch.Tensor:
    """One 3x3 layer (float32 OIHW, cout % 64 == 0, cin % 16 == 0) in the f32s row format, zero bias -> CPU uint8 blob."""
    w = np.ascontiguousarray(w.detach().cpu().numpy() if torch.is_tensor(w) else w, dtype=np.float32)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    assert w.shape[2:] == (3, 3)
    nbytes = lib().nqa_packed_conv_split_bytes(cout, cin)
    if not nbytes:
        raise ValueError(f"pack_conv_split: unsupported layer shape {w.shape}")
    blob = torch.empty(nbytes, dtype=torch.uint8)
    check(lib().nqa_pack_conv_split(w.ctypes.data, cout, cin, blob.data_ptr()))
    return blob


def conv3x3_split(inp: torch.Tensor, blob: torch.Tensor, cout: int, relu: bool = False) -> torch.Tensor:
    """split16 NHWC (n,H,W,cin) -> float32 NHWC (n,H,W,cout) through a layer packed by pack_conv_split."""
    dev = _need_cuda(inp, blob)
    assert inp.dtype == torch.float32 and inp.is_contiguous()
    n, h, w, cin = inp.shape
    out = torch.empty((n, h, w, cout), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_conv3x3_split, ptr(inp), n, h, w, cin, cout, ptr(blob), int(relu), ptr(out), stream_ptr(dev))
    return out


def relu_mask_split16(g: torch.Tensor, act: torch.Tensor, act_is_split16: bool) -> torch.Tensor:
    """g * (act > 0) as split16 records (float32-typed tensor of g's shape); act: float map or split16 records."""
    dev = _need_cuda(g, act)
    g = _f32c(g)
    assert act.dtype == torch.float32 and act.is_contiguous() and act.shape == g.shape
    c = g.shape[-1]
    out = torch.empty_like(g)
    _call(dev, lib().nqa_relu_mask_split16, ptr(g), ptr(act), int(act_is_split16), g.numel() // c, c, ptr(out),
          stream_ptr(dev))
    return out


def l2pool_backward(tap: torch.Tensor, pooled_split16: torch.Tensor, g_pooled: torch.Tensor, g_tap: torch.Tensor) -> None:
    """g_tap += d(L2-pool)/d(tap) applied to g_pooled; tap, g_tap float (n,H,W,C); pooled_split16, g_pooled (n,Ho,Wo,C)."""
    dev = _need_cuda(tap, pooled_split16, g_pooled, g_tap)
    n, h, w, c = tap.shape
    assert g_tap.shape == tap.shape and g_tap.is_contiguous() and tap.is_contiguous() and tap.dtype == torch.float32
    assert g_pooled.shape == pooled_split16.shape == (n, (h + 1) // 2, (w + 1) // 2, c)
    _call(dev, lib().nqa_l2pool_backward, ptr(tap), ptr(pooled_split16), ptr(_f32c(g_pooled)), n, h, w, c, ptr(g_tap),
          stream_ptr(dev))


def conv1_1_backward(gm: torch.Tensor, w0: torch.Tensor) -> torch.Tensor:
    """g * (relu1_1 > 0), float NHWC (n,H,W,64) -> gradient of the raw image, float NCHW (n,3,H,W); w0: conv1_1's OIHW weights."""
    dev = _need_cuda(gm, w0)
    gm, w0 = _f32c(gm), _f32c(w0)
    n, h, w, c = gm.shape
    assert c == 64 and tuple(w0.shape) == (64, 3, 3, 3)
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
    _call(dev, lib().nqa_conv1_1_backward, ptr(gm), ptr(w0), n, h, w, ptr(out), stream_ptr(dev))
    return out
