"""Deterministic shape fuzz: many odd sizes through both metrics against the CPU oracle.

Sizes are drawn once from a fixed seed so that every tile shape sees ragged edges, single rows and columns,
maps narrower than a tile, and stage-5 maps of 1..6 pixels; the oracle needs ~0.1 s per case at these sizes.
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(20240607)
SHAPES = sorted({(int(h), int(w)) for h, w in zip(_rng.integers(1, 97, 28), _rng.integers(1, 131, 28))} |
                {(1, 1), (2, 3), (31, 33), (32, 32), (33, 31), (64, 16), (16, 64), (47, 129), (96, 8)})


@pytest.fixture(scope="module")
def models():
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch import DISTS
    dev = torch.device("cuda:0")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return {"f16": DISTS(precision="f16").to(dev).eval(), "f32s": DISTS(precision="f32s").to(dev).eval(),
                "f32m": DISTS(precision="f32m").to(dev).eval(), "f32m2": DISTS(precision="f32m2").to(dev).eval(),
                "f32m4": DISTS(precision="f32m4").to(dev).eval(), "f16w": DISTS(precision="f16w").to(dev).eval(),
                "bf16": DISTS(precision="bf16").to(dev).eval(), "a32s": ADISTS(precision="f32s").to(dev).eval(),
                "a_auto": ADISTS().to(dev).eval()}


@pytest.mark.parametrize("h,w", SHAPES, ids=[f"{h}x{w}" for h, w in SHAPES])
def test_random_shape(h, w, models, oracle_convs):
    from nerf_qa_amd import synth
    from oracle import adists_oracle, dists_oracle
    dev = torch.device("cuda:0")
    b = 1 + (h * 7 + w) % 3
    kinds = [synth.KINDS[(h + w + i) % 4] for i in range(b)]
    xn, yn = synth.frame_batch([1000 + h * 131 + w + i for i in range(b)], h, w, kinds)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    m = models["f32s"]
    ref = dists_oracle.dists(x, y, oracle_convs, m.alpha.detach().cpu(), m.beta.detach().cpu())
    aref = adists_oracle.adists(x, y, oracle_convs)
    with torch.no_grad():
        # (f32s: 1e-7 .. 1e-6 on these frames; one 83x66 pair sits at 5-6e-6 -- tiny channel means, and whichever conv2_1
        # kernel runs moves it by a few 1e-6 although both are within 4e-7 of a float64 convolution,
        # tools/gpu_regw_split_check.py)
        for key, tol in (("f32s", 1e-5), ("f32m2", 1e-4), ("f32m", 1e-4), ("f32m4", 1e-4), ("f16w", 1e-4), ("f16", 1e-4), ("bf16", 1e-3)):
            got = models[key](x.to(dev), y.to(dev)).cpu()
            assert got.shape == ref.shape and (got - ref).abs().max().item() <= tol, (key, h, w, got, ref)
        got = models["a32s"](x.to(dev), y.to(dev), as_loss=False).cpu()
        # A-DISTS is discontinuous at dead channels: a knife-edge flip is ~5e-4, anything else ~1e-6
        # (a 1x1 frame is NaN in the reference itself -- 0/0 in the entropy weights -- and NaN here)
        assert torch.equal(torch.isnan(got), torch.isnan(aref)), ("a32s", h, w, got, aref)
        ok = ~torch.isnan(aref)
        assert not ok.any() or (got[ok] - aref[ok]).abs().max().item() <= 2e-5, ("a32s", h, w, got, aref)
        # the shipped default (exact f32 below 128x128 pixels, f32s above)
        got = models["a_auto"](x.to(dev), y.to(dev), as_loss=False).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(aref)), ("a_auto", h, w, got, aref)
        assert not ok.any() or (got[ok] - aref[ok]).abs().max().item() <= 2e-5, ("a_auto", h, w, got, aref)


class _GuardedWorkspace:
    """ops.Workspace stand-in: hands out exactly the bytes asked for, fenced by guard bytes on both sides."""
    GUARD = 1 << 16

    def __init__(self):
        self.full = None
        self.n = 0

    def get(self, nbytes, dev):
        self.n = max(int(nbytes), 256)
        self.full = torch.full((self.n + 2 * self.GUARD,), 0xA5, dtype=torch.uint8, device=dev)
        return self.full[self.GUARD:self.GUARD + self.n]

    def intact(self):
        g = self.GUARD
        return bool((self.full[:g] == 0xA5).all() and (self.full[g + self.n:] == 0xA5).all())


@pytest.mark.parametrize("prec", ["f16", "f32s", "f32m", "f32m2", "f32m4", "f16w"])
@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (1, 12), (3, 2), (7, 9), (17, 40), (33, 2), (64, 48), (97, 131)],
                         ids=lambda v: str(v))
def test_no_write_outside_the_workspace(h, w, prec, np_convs):
    """Every fused entry point with a fenced workspace and fenced outputs: the fences must survive."""
    from nerf_qa_amd import ops, synth
    dev = torch.device("cuda:0")
    packed = ops.pack_vgg_weights(np_convs, prec).to(dev)
    xn, yn = synth.frame_batch([5, 6], h, w)
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    calls = [lambda ws: ops.dists_forward(x, y, packed, prec, ws), lambda ws: ops.vgg_pyramid(x, packed, prec, ws)]
    if prec not in ("f32m", "f32m2", "f32m4", "f16w"):  # (DISTS modes: A-DISTS refuses them)
        calls.append(lambda ws: ops.adists_forward(x, y, packed, prec, ws, with_map=True))
    for call in calls:
        ws = _GuardedWorkspace()
        call(ws)
        torch.cuda.synchronize()
        assert ws.intact(), f"{h}x{w} {prec}: a kernel wrote outside its workspace"


_rng2 = np.random.default_rng(77)
OP_CASES = [(int(l), int(n), int(h), int(w)) for l, n, h, w in zip(
    _rng2.choice([1, 2, 3, 4, 6, 7, 9, 10, 12], 24), _rng2.integers(1, 4, 24), _rng2.integers(1, 41, 24),
    _rng2.integers(1, 71, 24))] + [(8, 1, 8, 16), (8, 1, 8, 17), (5, 2, 9, 32), (5, 2, 9, 33), (3, 1, 4, 15), (12, 2, 1, 1)]


@pytest.mark.parametrize("prec", ["f16", "bf16", "f32", "f32s"])
@pytest.mark.parametrize("layer,n,h,w", OP_CASES, ids=[f"L{c[0]}-{c[1]}x{c[2]}x{c[3]}" for c in OP_CASES])
def test_random_conv_layer(layer, n, h, w, prec, np_convs):
    """One conv layer at a random ragged size (every tile variant: 16- and 32-wide, 4- and 8-wave) vs F.conv2d."""
    import torch.nn.functional as F
    from nerf_qa_amd import ops, synth
    dev = torch.device("cuda:0")
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}.get(prec, torch.float32)
    rtol = {"f16": 1.2e-3, "bf16": 9e-3}.get(prec, 2e-5)
    cin = ops.CONV_CIN[layer]
    a = torch.from_numpy((synth.uniform(layer * 1000 + h * 71 + w, n * h * w * cin) * 4 - 2).astype(np.float32)
                         .reshape(n, h, w, cin)).clamp_min(0).to(dt)
    wq = torch.from_numpy(np_convs[layer][0]).to(dt).float()
    ref = F.relu(F.conv2d(a.float().permute(0, 3, 1, 2), wq, torch.from_numpy(np_convs[layer][1]), padding=1))
    packed = ops.pack_vgg_weights(np_convs, prec).to(dev)
    inp = ops.split16_encode(a.to(dev)) if prec == "f32s" else a.to(dev)
    for variant in (0, 1, 2, 1 + 16, 1 + 32):  # +16: the round-1 forms (conv2_1 on the implicit GEMM); +32: register weights on Cin = 128
        ops.set_conv_variant(variant)
        try:
            out = ops.conv3x3_relu(inp, layer, packed, prec)
        finally:
            ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
        got = (ops.split16_decode(out) if prec == "f32s" and layer not in ops.TAP_LAYERS else out.float())
        got = got.permute(0, 3, 1, 2).cpu()
        err = (got - ref).abs().max().item()
        assert err <= rtol * (ref.abs().max().item() + 1e-30), (layer, n, h, w, prec, variant, err)


@pytest.mark.parametrize("h,w", [(21, 21), (22, 45), (37, 53), (64, 80), (20, 50), (43, 21), (90, 97)], ids=lambda v: str(v))
def test_random_adists_map(h, w, models, oracle_convs):
    """as_map=True at sizes around the 21-pixel window switch (stage 0 windowed or not, later stages global)."""
    from nerf_qa_amd import synth
    from oracle import adists_oracle
    dev = torch.device("cuda:0")
    xn, yn = synth.frame_batch([h, w], h, w, ["blur", "noise10"])
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    ref = adists_oracle.adists(x, y, oracle_convs, as_map=True)
    with torch.no_grad():
        got = models["a32s"](x.to(dev), y.to(dev), as_map=True).cpu()
    assert got.shape == ref.shape == (2, 2, h, w)
    # 21x21: the stage-0 window map is a single value whose unbiased std is NaN in the reference -- and here
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    ok = ~torch.isnan(ref)
    assert not ok.any() or (got[ok] - ref[ok]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("prec", ["f16", "f32", "f32s"])
@pytest.mark.parametrize("n,h,w,c", [(1, 1, 1, 64), (2, 3, 2, 128), (1, 5, 33, 64), (3, 2, 7, 512), (1, 31, 1, 256),
                                     (2, 18, 19, 128)], ids=lambda v: str(v))
def test_random_l2pool(n, h, w, c, prec):
    from nerf_qa_amd import ops, synth
    from oracle import dists_oracle
    dev = torch.device("cuda:0")
    dt = torch.float16 if prec == "f16" else torch.float32
    a = torch.from_numpy((synth.uniform(n * 7 + h * 13 + w, n * h * w * c) * 3).astype(np.float32).reshape(n, h, w, c)).to(dt)
    ref = dists_oracle.l2pool(a.float().permute(0, 3, 1, 2))
    out = ops.l2pool(a.to(dev), prec)
    got = (ops.split16_decode(out) if prec == "f32s" else out.float()).permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= (1.2e-3 if prec == "f16" else 2e-5) * ref.abs().max().item()


@pytest.mark.parametrize("h,w,b", [(1, 1, 1), (2, 3, 3), (9, 4, 2), (40, 70, 1), (130, 67, 2)], ids=lambda v: str(v))
def test_random_from_feats(h, w, b):
    """forward_from_feats' statistics on caller-provided float NCHW pyramids of odd sizes (stats_nchw_kernel)."""
    from nerf_qa_amd import ops, synth
    from oracle import dists_oracle
    dev = torch.device("cuda:0")
    chns = (3, 64, 128, 256, 512, 512)
    dims, hh, ww = [(h, w)], h, w
    for k in range(5):
        if k:
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        dims.append((hh, ww))
    mk = lambda seed, shape: torch.from_numpy((synth.uniform(seed, int(np.prod(shape))) * 2).astype(np.float32).reshape(shape))
    f0 = [mk(50 + k, (b, c, dh, dw)) for k, (c, (dh, dw)) in enumerate(zip(chns, dims))]
    f1 = [(f + 0.25 * (mk(80 + k, tuple(f.shape)) - 1)).clamp_min(0) for k, f in enumerate(f0)]
    # the oracle in float64: with 1-2 pixel maps a channel's variance can sit at 1e-5, where the reference's own
    # float32 two-pass variance is only good to 1e-2 relative; the kernel (fp64 sums) is compared with the exact value
    r1, r2 = dists_oracle.dists_stats([f.double() for f in f0], [f.double() for f in f1])
    r1, r2 = r1.float(), r2.float()
    s1, s2 = ops.dists_stats_nchw([f.to(dev) for f in f0], [f.to(dev) for f in f1])
    e1, e2 = (s1.cpu() - r1).abs(), (s2.cpu() - r2).abs()
    assert e1.max().item() <= 2e-6 and e2.max().item() <= 2e-5, (e1.max().item(), e2.max().item(), int(e2.argmax()) % 1475)


_rng3 = np.random.default_rng(5)
RESIZE_CASES = [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(
    _rng3.integers(1, 400, 20), _rng3.integers(1, 500, 20), _rng3.integers(1, 300, 20), _rng3.integers(1, 300, 20))] + \
    [(1, 1, 7, 5), (2, 2, 1, 1), (1080, 1920, 256, 256), (300, 1, 10, 1)]


@pytest.mark.parametrize("hin,win,hout,wout", RESIZE_CASES, ids=[f"{a}x{b}-{c}x{d}" for a, b, c, d in RESIZE_CASES])
def test_random_resize(hin, win, hout, wout):
    """Random up- and down-scales: bit-exact with the installed Pillow, 1e-6 from F.interpolate."""
    import torch.nn.functional as F
    from PIL import Image
    from nerf_qa_amd import ops, synth
    dev = torch.device("cuda:0")
    img = (synth.uniform(hin * 1000 + win, hin * win * 3) * 256).astype(np.uint8).reshape(1, hin, win, 3)
    d = torch.from_numpy(img).to(dev)
    got = ops.resize_pil_bilinear_u8(d, (hout, wout)).cpu().numpy()[0]
    want = np.asarray(Image.fromarray(img[0]).resize((wout, hout), Image.BILINEAR))
    assert np.array_equal(got, want), np.abs(got.astype(int) - want).max()
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float() / 255.0
    ref = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=False)
    assert (ops.u8_resize_bilinear_f32(d, (hout, wout)).cpu() - ref).abs().max().item() <= 1e-6
