"""Operator-level parity of the HIP kernels against the CPU oracle (GPU box only).

Every call goes through the C ABI (nerf_qa_amd.ops -> libnqa_hip.so).  Tolerances:
  f32  : accumulation-order differences only            -> 2e-5 relative to the map's scale
  f32s : split-f16 products (hi*hi + hi*lo + lo*hi), ~2^-21 per product -> same bar as f32
  f16  : + one rounding of each stored activation (2^-11)
  bf16 : + one rounding of each stored activation (2^-8)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PRECS = ["f32", "f32s", "f16", "bf16"]
DT = {"f32": torch.float32, "f32s": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
OUT_RTOL = {"f32": 2e-5, "f32s": 2e-5, "f16": 1.2e-3, "bf16": 9e-3}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def packed(np_convs, dev):
    from nerf_qa_amd import ops
    return {p: ops.pack_vgg_weights(np_convs, p).to(dev) for p in PRECS}


def _rand(shape, seed, lo=0.0, hi=1.0):
    from nerf_qa_amd import synth
    n = int(np.prod(shape))
    return torch.from_numpy((synth.uniform(seed, n) * (hi - lo) + lo).astype(np.float32).reshape(shape))


def _floats(out, prec, split):
    """Activation tensor -> float32; in f32s the maps between conv layers are split16-encoded."""
    from nerf_qa_amd import ops
    return ops.split16_decode(out) if (prec == "f32s" and split) else out.float()


def _close(got, ref, rtol, what):
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (rtol {rtol})"
    return err / scale


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("shape", [(2, 3, 37, 53), (1, 3, 16, 16), (3, 3, 5, 70)])
def test_conv1_1(prec, shape, np_convs, packed, dev):
    from nerf_qa_amd import ops
    x = _rand(shape, 11)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    w, b = torch.from_numpy(np_convs[0][0]), torch.from_numpy(np_convs[0][1])
    ref = F.relu(F.conv2d((x - mean) / std, w, b, padding=1))
    out = ops.conv1_1(x.to(dev), packed[prec], prec)
    assert out.dtype == DT[prec] and out.shape == (shape[0], shape[2], shape[3], 64)
    got = _floats(out, prec, True).permute(0, 3, 1, 2).cpu()
    _close(got, ref, OUT_RTOL[prec], f"conv1_1[{prec}]")


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 3, 37, 53), (1, 3, 16, 16), (3, 3, 5, 70), (1, 3, 64, 96), (2, 3, 9, 33)])
@pytest.mark.parametrize("stage1", [0, 1], ids=["persistent", "tile"])
def test_conv1_fused(stage1, prec, shape, np_convs, packed, dev):
    """Stage 1 in one kernel (conv1_1 on MFMA feeding conv1_2 through LDS) == the two oracle convs.

    conv1_1's inputs and weights are rounded to 16 bits here (they are fp32 in the two-kernel
    path), so the bound is the two-layer accumulation of 16-bit rounding, not a single rounding.
    """
    from nerf_qa_amd import ops
    x = _rand(shape, 21)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    w0, b0 = torch.from_numpy(np_convs[0][0]), torch.from_numpy(np_convs[0][1])
    w1, b1 = torch.from_numpy(np_convs[1][0]), torch.from_numpy(np_convs[1][1])
    ref = F.relu(F.conv2d(F.relu(F.conv2d((x - mean) / std, w0, b0, padding=1)), w1, b1, padding=1))
    ops.set_conv_variant(1 | (stage1 << 2))  # both stage-1 kernels (include/nqa.h, nqa_set_conv_variant)
    try:
        out = ops.conv1_fused(x.to(dev), packed[prec], prec)
    finally:
        ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
    assert out.dtype == DT[prec] and out.shape == (shape[0], shape[2], shape[3], 64)
    got = out.float().permute(0, 3, 1, 2).cpu()
    _close(got, ref, 3 * OUT_RTOL[prec], f"conv1_fused[{prec}] {shape}")


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 3, 4, 32), (1, 3, 9, 33), (3, 3, 37, 70), (1, 3, 64, 100),
                                   (1, 3, 130, 95), (2, 3, 1, 1), (1, 3, 3, 200)], ids=lambda s: "x".join(map(str, s)))
def test_f32s_stage1_fused(shape, np_convs, packed, dev):
    """f32s stage 1 in one kernel (conv1_regw_split_kernel: conv1_1 on MFMA with three-term products feeding a three-term
    conv1_2 through LDS, 4 x 32 tiles) == the two convolutions in float64, to float32-class accuracy, at ragged sizes
    (partial tiles in both directions, frames smaller than a tile, more tiles than CUs), and == the round-2 pair of
    kernels (first-forms bit of nqa_set_conv_variant) to the rounding of another summation order."""
    from nerf_qa_amd import ops
    x = _rand(shape, 77)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64).view(1, 3, 1, 1)
    w0, b0 = torch.from_numpy(np_convs[0][0]).double(), torch.from_numpy(np_convs[0][1]).double()
    w1, b1 = torch.from_numpy(np_convs[1][0]).double(), torch.from_numpy(np_convs[1][1]).double()
    ref = F.relu(F.conv2d(F.relu(F.conv2d((x.double() - mean) / std, w0, b0, padding=1)), w1, b1, padding=1))
    got = {}
    for name, variant in (("fused", ops.DEFAULT_CONV_VARIANT), ("pair", ops.DEFAULT_CONV_VARIANT | 16)):
        ops.set_conv_variant(variant)
        try:
            tap1 = ops.vgg_pyramid(x.to(dev), packed["f32s"], "f32s")[0]
        finally:
            ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
        assert tap1.dtype == torch.float32 and tap1.shape == (shape[0], shape[2], shape[3], 64)
        got[name] = tap1.permute(0, 3, 1, 2).double().cpu()
        assert torch.isfinite(got[name]).all()
    op = ops.conv1_fused(x.to(dev), packed["f32s"], "f32s")  # the single-operator entry point: the same kernel
    assert op.dtype == torch.float32 and torch.equal(op.permute(0, 3, 1, 2).double().cpu(), got["fused"])
    scale = ref.abs().max().item()
    e_f, e_p = (got["fused"] - ref).abs().max().item() / scale, (got["pair"] - ref).abs().max().item() / scale
    print(f"\n f32s stage 1 {shape}: fused {e_f:.2e}, two kernels {e_p:.2e} of the largest activation")
    assert e_f <= 2e-6 and e_p <= 2e-6, (shape, e_f, e_p)


def test_conv1_fused_rejects_f32(packed, dev):
    from nerf_qa_amd import NqaError, ops
    with pytest.raises(NqaError):
        ops.conv1_fused(torch.zeros(1, 3, 8, 8, device=dev), packed["f32"], "f32")


def test_pool_stats_matches_separate_kernels(packed, alpha_beta, dev):
    """The fused pool+statistics pass (taps 1..4 of the DISTS path) against the stand-alone L2-pool:
    run the pyramid op by op with nqa_l2pool and compare the taps with nqa_vgg_pyramid's."""
    from nerf_qa_amd import ops, synth
    x, _ = synth.frame_batch([3], 45, 70)
    x = torch.from_numpy(x).to(dev)
    prec = "f16"
    taps = ops.vgg_pyramid(x, packed[prec], prec)
    h = ops.conv1_fused(x, packed[prec], prec)
    layer = 2
    for k in range(5):
        assert torch.equal(h, taps[k]), f"tap {k + 1} differs"
        if k == 4:
            break
        h = ops.l2pool(h, prec)
        for _ in range((2, 2, 3, 3, 3)[k + 1]):
            h = ops.conv3x3_relu(h, layer, packed[prec], prec)
            layer += 1


CONV_CASES = [  # (layer, n, H, W)
    (1, 2, 13, 37), (1, 1, 32, 64), (1, 2, 9, 16),
    (2, 2, 11, 40), (3, 1, 8, 33), (3, 3, 16, 16), (4, 1, 7, 7),
    (6, 2, 12, 35), (7, 1, 6, 18), (9, 2, 9, 20), (12, 3, 4, 16), (12, 1, 2, 2),
]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("layer,n,h,w", CONV_CASES)
def test_conv3x3_relu(prec, layer, n, h, w, np_convs, packed, dev):
    from nerf_qa_amd import ops
    cin = ops.CONV_CIN[layer]
    # post-ReLU-like input: half zeros, rest in [0, 2); stored in the activation dtype
    a = _rand((n, h, w, cin), 100 + layer, -2.0, 2.0).clamp_min(0).to(DT[prec])
    wq = torch.from_numpy(np_convs[layer][0]).to(DT[prec]).float()
    b = torch.from_numpy(np_convs[layer][1])
    ref = F.relu(F.conv2d(a.float().permute(0, 3, 1, 2), wq, b, padding=1))
    inp = ops.split16_encode(a.to(dev)) if prec == "f32s" else a.to(dev)
    out = ops.conv3x3_relu(inp, layer, packed[prec], prec)
    got = _floats(out, prec, layer not in ops.TAP_LAYERS).permute(0, 3, 1, 2).cpu()
    _close(got, ref, OUT_RTOL[prec], f"conv layer {layer} [{prec}] {n}x{h}x{w}")


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("n,h,w,c", [(2, 9, 13, 64), (1, 16, 16, 128), (2, 1, 5, 256), (1, 7, 2, 512)])
def test_l2pool(prec, n, h, w, c, dev):
    from nerf_qa_amd import ops
    from oracle import dists_oracle
    a = _rand((n, h, w, c), 7, 0.0, 3.0).to(DT[prec])
    ref = dists_oracle.l2pool(a.float().permute(0, 3, 1, 2))
    out = ops.l2pool(a.to(dev), prec)
    assert out.shape == (n, (h + 1) // 2, (w + 1) // 2, c)
    _close(_floats(out, prec, True).permute(0, 3, 1, 2).cpu(), ref, OUT_RTOL[prec], f"l2pool[{prec}]")


@pytest.mark.parametrize("prec", PRECS)
def test_nhwc_to_nchw(prec, dev):
    from nerf_qa_amd import ops
    a = _rand((2, 5, 9, 72), 3).to(DT[prec])
    out = ops.nhwc_to_nchw_f32(a.to(dev), prec).cpu()
    assert torch.equal(out, a.float().permute(0, 3, 1, 2))


def test_split16_roundtrip(dev):
    """split16 = (half hi, half lo) per element: decode(encode(v)) is v to ~2^-21, exact for half-representable v."""
    from nerf_qa_amd import ops
    a = _rand((2, 5, 7, 64), 9, 0.0, 300.0)
    a[0, 0, 0, :8] = torch.tensor([0.0, 1.0, 0.5, 1024.0, 0.75, 65504.0, 2.0 ** -14, 0.333251953125])
    enc = ops.split16_encode(a.to(dev))
    back = ops.split16_decode(enc).cpu()
    assert (back - a).abs().max().item() <= 2.0 ** -21 * 300.0
    assert torch.equal(back[0, 0, 0, :8], a[0, 0, 0, :8])
    raw = enc.cpu().view(torch.float16).view(2, 5, 7, 4, 4, 8)  # [..., group of 16 ch, (hi0 hi1 lo0 lo1), 8]
    hi = raw[..., 0:2, :].reshape(2, 5, 7, 4, 16).reshape(2, 5, 7, 64)
    assert torch.equal(hi, a.half())


def test_stats_nchw_and_score(alpha_beta, dev):
    """forward_from_feats path: statistics on caller-provided NCHW maps + fused score."""
    from nerf_qa_amd import ops
    from oracle import dists_oracle
    chns = (3, 64, 128, 256, 512, 512)
    dims = [(33, 47), (33, 47), (17, 24), (9, 12), (5, 6), (3, 3)]
    f0 = [_rand((2, c, h, w), 20 + k, 0, 2) for k, (c, (h, w)) in enumerate(zip(chns, dims))]
    f1 = [(f + 0.3 * _rand(tuple(f.shape), 40 + k, -1, 1)).clamp_min(0) for k, f in enumerate(f0)]
    f0[3][:, 5] = 0.0  # a dead channel on both sides: S2 must come out as exactly 1
    f1[3][:, 5] = 0.0
    r1, r2 = dists_oracle.dists_stats(f0, f1)
    s1, s2 = ops.dists_stats_nchw([f.to(dev) for f in f0], [f.to(dev) for f in f1])
    assert (s1.cpu() - r1).abs().max().item() < 2e-6
    assert (s2.cpu() - r2).abs().max().item() < 2e-5
    assert s2[0, 3 + 64 + 128 + 5].item() == 1.0
    alpha, beta = alpha_beta
    ref = dists_oracle.dists_score(r1, r2, alpha, beta)
    got = ops.dists_score(s1, s2, alpha.to(dev), beta.to(dev)).cpu()
    assert (got - ref).abs().max().item() < 2e-6
