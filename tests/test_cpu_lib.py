"""CPU-side checks: the C-ABI library loads, exports every declared symbol, validates
arguments, and its host-side weight packer lays tiles out as the kernels expect."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from nerf_qa_amd import build, _lib
    build.build()
    return _lib.lib()


def test_exports_match_header(lib):
    from nerf_qa_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nqa.h")).read()
    declared = set(re.findall(r"\b(nqa_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)


def test_argument_errors(lib):
    assert lib.nqa_conv1_1(None, 1, 8, 8, None, 0, None, None) == -1
    assert b"null" in lib.nqa_last_error()
    assert lib.nqa_packed_weights_bytes(0) > lib.nqa_packed_weights_bytes(1) == lib.nqa_packed_weights_bytes(2)
    assert lib.nqa_workspace_bytes(0, 8, 8, 0) == 0


def _unpack_layer(blob, off, cin, cout, dtype, cpc, m16=False):
    """Invert the documented tile layout back to OIHW (numpy).  m16: the chunk swizzle of the 16x16x32-MFMA
    layers (16-bit modes, layers 2..13): 2*((n>>2)&1) instead of (n>>2)&3."""
    kc, bn = 4 * cpc, 64
    ncc = cin // kc
    n_el = cin * cout * 9
    raw = blob[off:off + n_el * dtype().itemsize].view(dtype).reshape(cout // bn, ncc, 9, bn, 4, cpc)
    w = np.zeros((cout, cin, 9), dtype=dtype)
    for n in range(bn):
        for pos in range(4):
            c = pos ^ (((n >> 2) & 1) * 2 if m16 else (n >> 2) & 3)
            # raw[ct, cc, t, n, pos, j] -> w[ct*bn+n, cc*kc + c*cpc + j, t]
            src = raw[:, :, :, n, pos, :]  # (ct, cc, t, j)
            for ct in range(cout // bn):
                for cc in range(ncc):
                    w[ct * bn + n, cc * kc + c * cpc:cc * kc + (c + 1) * cpc, :] = src[ct, cc].T
    return w.reshape(cout, cin, 3, 3)


@pytest.mark.parametrize("prec", ["f32", "bf16", "f16"])
def test_pack_roundtrip(prec, np_convs, lib):
    from nerf_qa_amd import ops
    blob = ops.pack_vgg_weights(np_convs, prec).numpy()
    # recompute the documented offsets
    esz = 4 if prec == "f32" else 2
    al = lambda v: (v + 255) // 256 * 256
    assert not blob[:256].any()  # zero page: the source of out-of-image halo pixels
    off = 256 + al(27 * 64 * 4 + 64 * 4) + 6144  # + conv1_1 as 16-bit MFMA fragments (fused stage-1 kernel)
    w0 = blob[256:256 + 27 * 64 * 4].view(np.float32).reshape(9, 3, 64)
    assert np.array_equal(w0, np_convs[0][0].reshape(64, 3, 9).transpose(2, 1, 0))
    wm = blob[256 + al(27 * 64 * 4 + 64 * 4):][:6144].view(np.uint16).reshape(3, 64, 2, 8)
    if prec == "f32":
        assert not wm.any()
    else:
        conv = (lambda a: a.astype(np.float16).view(np.uint16)) if prec == "f16" else \
            (lambda a: torch.from_numpy(a.copy()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))
        w00 = np_convs[0][0]  # (64, 3, 3, 3) = [cout][c][ky][kx]
        for ky in range(3):
            for hh in range(2):
                for j in range(8):
                    kx, c = 2 * hh + j // 4, j % 4
                    want = conv(w00[:, c, ky, kx]) if (kx < 3 and c < 3) else np.zeros(64, np.uint16)
                    assert np.array_equal(wm[ky, :, hh, j], want)
    for l in (1, 2, 7, 12):
        cin, cout = ops.CONV_CIN[l], ops.CONV_COUT[l]
        o = off + sum(al(ops.CONV_CIN[i] * ops.CONV_COUT[i] * 9 * esz) + al(ops.CONV_COUT[i] * 4 + 4) for i in range(1, l))
        w = np_convs[l][0]
        if prec == "f32":
            got = _unpack_layer(blob, o, cin, cout, np.float32, 4)
            assert np.array_equal(got, w)
        elif prec == "f16":
            got = _unpack_layer(blob, o, cin, cout, np.float16, 8, m16=l >= 2)
            assert np.array_equal(got, w.astype(np.float16))
        else:
            got = _unpack_layer(blob, o, cin, cout, np.uint16, 8, m16=l >= 2)
            ref = torch.from_numpy(w).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
            assert np.array_equal(got, ref)
        bias = blob[o + al(cin * cout * 9 * esz):][:cout * 4 + 4].view(np.float32)
        assert np.array_equal(bias[:cout], np_convs[l][1]) and bias[cout] == 1.0  # no weight scale in these modes


def test_pack_split_f16(np_convs, lib):
    """f32s blob: every weight, times the layer's power-of-two scale s (largest |w|*s in [512, 1024): hi and lo both
    NORMAL halves), as an f16 (hi, lo) pair, rows [hi0-7|hi8-15|lo0-7|lo8-15]; 1/s follows the bias.
    (hi + lo)/s == w to 2^-22 relative -- float32-class, where unscaled weights (~1e-2, lo subnormal) reach 2^-19."""
    from nerf_qa_amd import ops
    blob = ops.pack_vgg_weights(np_convs, "f32s").numpy()
    # (behind the 13 layers: conv1_2 and, since round 4, conv2_1 as two-term register fragments -- conv1_regw_split_kernel,
    # conv3x3_regw_split_kernel -- + conv1_1 as (hi, lo) MFMA fragments for the fused f32s stage-1 kernel)
    n1, n2 = 64 * 64 * 9 * 2 * 2, 128 * 64 * 9 * 2 * 2
    tail = n1 + n2 + 2 * 4 * 2 * 64 * 16
    assert blob.nbytes == ops.pack_vgg_weights(np_convs, "f32").numpy().nbytes + tail
    # those fragments: element j of lane (l15, c4) of k-step ks = tap ks % 9 of input channel 32*(ks//9) + 8*c4 + j, output
    # channel 16*g + l15, times the layer's scale, as hi / lo halves
    al0 = lambda v: (v + 255) // 256 * 256
    o1 = 256 + al0(27 * 64 * 4 + 64 * 4) + 6144
    o2 = o1 + al0(64 * 64 * 36) + al0(64 * 4 + 4)
    for layer, cout, frag_off, nbytes, bias_off in ((1, 64, blob.nbytes - tail, n1, o1 + al0(64 * 64 * 36)),
                                                    (2, 128, blob.nbytes - tail + n1, n2, o2 + al0(64 * 128 * 36))):
        frag = blob[frag_off:][:nbytes].view(np.float16).reshape(cout // 16, 2, 18, 64, 8)
        wl = np_convs[layer][0].reshape(cout, 64, 9)
        sc = 1.0 / float(blob[bias_off:][:(cout + 1) * 4].view(np.float32)[cout])
        for g, ks, lane in ((0, 0, 0), (cout // 16 - 1, 17, 63), (2, 9, 37), (1, 4, 16)):
            co, ci0, t = 16 * g + (lane & 15), 32 * (ks // 9) + 8 * (lane >> 4), ks % 9
            v = (wl[co, ci0:ci0 + 8, t] * np.float32(sc)).astype(np.float32)
            assert np.array_equal(frag[g, 0, ks, lane], v.astype(np.float16))
            assert np.array_equal(frag[g, 1, ks, lane], (v - v.astype(np.float16).astype(np.float32)).astype(np.float16))
    al = lambda v: (v + 255) // 256 * 256
    off = 256 + al(27 * 64 * 4 + 64 * 4) + 6144
    for l in (1, 4, 12):
        cin, cout = ops.CONV_CIN[l], ops.CONV_COUT[l]
        o = off + sum(al(ops.CONV_CIN[i] * ops.CONV_COUT[i] * 9 * 4) + al(ops.CONV_COUT[i] * 4 + 4) for i in range(1, l))
        ncc = cin // 16
        t = blob[o:o + cin * cout * 36].view(np.float16).reshape(cout // 64, ncc, 9, 64, 4, 8)
        w = np_convs[l][0].reshape(cout, cin, 9)
        hi = np.zeros((cout, cin, 9), np.float16)
        lo = np.zeros((cout, cin, 9), np.float16)
        for n in range(64):
            for pos in range(4):
                c = pos ^ ((n >> 2) & 3)
                dst = hi if c < 2 else lo
                for cc in range(ncc):
                    ch = cc * 16 + (c & 1) * 8
                    dst[n::64, ch:ch + 8, :] = t[:, cc, :, n, pos, :].transpose(0, 2, 1)
        bias = blob[o + al(cin * cout * 36):][:cout * 4 + 4].view(np.float32)
        assert np.array_equal(bias[:cout], np_convs[l][1])
        inv = float(bias[cout])
        scale = 1.0 / inv
        assert scale == 2.0 ** round(np.log2(scale)) and 512 <= np.abs(w).max() * scale < 1024
        ws = w * np.float32(scale)
        assert np.array_equal(hi, ws.astype(np.float16))
        assert np.array_equal(lo, (ws - hi.astype(np.float32)).astype(np.float16))
        rec = (hi.astype(np.float64) + lo.astype(np.float64)) * inv
        big = np.abs(w) > 2.0 ** -10 * np.abs(w).max()  # (weights a thousand times below the largest: absolute bound)
        assert (np.abs(rec - w)[big] <= 2.0 ** -21 * np.abs(w)[big]).all()
        assert np.abs(rec - w).max() <= 2.0 ** -22 * np.abs(w).max()


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (1, 12), (3, 2), (7, 7), (64, 64)])
def test_workspace_covers_the_largest_stage(h, w, lib):
    """Below ~8 pixels a side a LATER stage's map is the largest (ceil(./2) pixels, doubling channels): the
    ping-pong buffers must be sized for it (a 1x1 frame: 64 elements at stage 1, 512 at stages 4 and 5)."""
    chans = (64, 128, 256, 512, 512)
    hk, wk, biggest = h, w, 0
    for c in chans:
        biggest = max(biggest, hk * wk * c)
        hk, wk = (hk + 1) // 2, (wk + 1) // 2
    for prec, esz in ((0, 4), (2, 2), (3, 4)):
        assert lib.nqa_workspace_bytes(4, h, w, prec) >= 2 * 4 * biggest * esz
        assert lib.nqa_adists_workspace_bytes(2, h, w, prec) >= 2 * 4 * biggest * esz


def test_no_cpu_fallback():
    """CPU tensors must be refused, not silently computed somewhere else."""
    from nerf_qa_amd import _lib, ops
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(_lib.NqaError):
        ops.dists_forward(x, x, torch.zeros(16, dtype=torch.uint8), "f32")


def test_tracked_kernel_resources_match_the_last_build():
    """nerf_qa_amd/kernel_resources.json is tracked and only rewritten on request (`python -m nerf_qa_amd.build --force
    --emit-resources`); a build in this tree leaves its own report in nerf_qa_amd/build/.  When that exists the two
    must agree -- i.e. the committed report describes the sources as they are -- and no hand-scheduled kernel spills."""
    import json
    from nerf_qa_amd import build as b
    tracked = json.load(open(b.RESOURCES))
    assert not b.check_no_scratch(tracked)
    if not os.path.exists(b.RESOURCES_BUILD) or os.path.getmtime(b.RESOURCES_BUILD) < max(
            os.path.getmtime(os.path.join(b.CSRC, f)) for f in os.listdir(b.CSRC) if f.endswith((".hip", ".h"))):
        pytest.skip("no report of a build of the current sources in this tree")
    fresh = json.load(open(b.RESOURCES_BUILD))
    assert fresh == tracked, sorted(k for k in set(fresh) | set(tracked) if fresh.get(k) != tracked.get(k))[:5]
