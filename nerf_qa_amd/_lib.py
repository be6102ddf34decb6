"""ctypes binding of libnqa_hip.so (include/nqa.h).

The library is the product: if it is missing or does not load, importing this module's
`lib()` raises.  There is no CPU or eager-PyTorch fallback anywhere in nerf_qa_amd.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to; must come first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NQA_LIB") or os.path.join(_HERE, "libnqa_hip.so")  # NQA_LIB: development builds only

PREC_F32, PREC_BF16, PREC_F16, PREC_F32S, PREC_F32M, PREC_F32M2, PREC_F32M4, PREC_F16W = 0, 1, 2, 3, 4, 5, 6, 7
PREC_NAMES = {"f32": PREC_F32, "fp32": PREC_F32, "bf16": PREC_BF16, "f16": PREC_F16, "fp16": PREC_F16,
              "f32s": PREC_F32S, "f32m": PREC_F32M, "f32m2": PREC_F32M2, "f32m4": PREC_F32M4, "f16w": PREC_F16W}
PREC_DTYPE = {PREC_F32: torch.float32, PREC_BF16: torch.bfloat16, PREC_F16: torch.float16, PREC_F32S: torch.float32}
# NQA_MIXED_STAGES: the mixed modes run their first pyramid stages as f16 kernels on two-term weights, the rest as f32s
MIXED_STAGES = {PREC_F32M: 3, PREC_F32M2: 2, PREC_F32M4: 4, PREC_F16W: 5}


def stage_prec(prec: int, stage: int) -> int:
    """Kernel precision of pyramid stage `stage` (0-based) / of tap stage+1 in mode `prec` (include/nqa.h)."""
    if prec not in MIXED_STAGES:
        return prec
    return PREC_F16 if stage < MIXED_STAGES[prec] else PREC_F32S
NUM_CONVS, NUM_TAPS, TOTAL_CHNS = 13, 6, 1475
K_NAMES = ("conv1_1", "conv_igemm", "l2pool", "stats", "adists", "prep", "pool_seam")

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_SIGNATURES = {
    "nqa_version": (_i, []),
    "nqa_last_error": (C.c_char_p, []),
    "nqa_packed_weights_bytes": (_sz, [_i]),
    "nqa_pack_vgg_weights": (_i, [C.POINTER(_vp), C.POINTER(_vp), _i, _vp]),
    "nqa_conv1_1": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "nqa_conv1_fused": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "nqa_conv3x3_relu": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "nqa_l2pool": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nqa_nhwc_to_nchw_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nqa_split16_encode": (_i, [_vp, C.c_long, _i, _vp, _vp]),
    "nqa_split16_decode": (_i, [_vp, C.c_long, _i, _vp, _vp]),
    "nqa_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nqa_vgg_pyramid": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _sz, C.POINTER(_vp), _vp]),
    "nqa_dists_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "nqa_stats_scratch_bytes": (_sz, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "nqa_dists_stats_nchw": (_i, [C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                  _vp, _sz, _vp, _vp, _vp]),
    "nqa_dists_score": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "nqa_conv_pool_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nqa_dists_fused_taps": (_i, [_i, _i, _i, _i, C.POINTER(_i)]),
    "nqa_conv_pool_stats": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "nqa_conv1_pool_stats": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "nqa_adists_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "nqa_adists_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "nqa_adists_forward_map": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "nqa_u8hwc_to_f32nchw": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "nqa_resize_bilinear_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nqa_u8_resize_bilinear_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nqa_resize_pil_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "nqa_resize_pil_bilinear_u8": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "nqa_packed_conv_split_bytes": (_sz, [_i, _i]),
    "nqa_pack_conv_split": (_i, [_vp, _i, _i, _vp]),
    "nqa_conv3x3_split": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "nqa_relu_mask_split16": (_i, [_vp, _vp, _i, C.c_long, _i, _vp, _vp]),
    "nqa_l2pool_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "nqa_conv1_1_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "nqa_set_conv_variant": (_i, [_i]),
    "nqa_timing_enable": (_i, [_i]),
    "nqa_timing_collect": (_i, [C.POINTER(_i), C.POINTER(C.c_double)]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class NqaError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libnqa_hip.so once; raise if it is not there (build with `python -m nerf_qa_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NqaError(f"{LIB_PATH} not found: the HIP library is required "
                           "(run `python -m nerf_qa_amd.build`); there is no fallback path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        if handle.nqa_version() != 1:
            raise NqaError("libnqa_hip.so: ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise NqaError(f"libnqa_hip error {rc}: {lib().nqa_last_error().decode()}")


def prec_id(prec) -> int:
    if isinstance(prec, str):
        return PREC_NAMES[prec.lower()]
    return int(prec)


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
