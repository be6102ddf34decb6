"""BASELINE.json's full-size configs against goldens frozen from the imported reference
(oracle/make_goldens.py `full`: nerf_qa.DISTS_pytorch.DISTS_pt.DISTS / nerf_qa.ADISTS.ADISTS run pair by pair on
CPU), at their STATED batch sizes and on two more VGG weight sets:

  configs[1]  B=32 of 256x256, DISTS            weight gains 1.0, 1.3, 1.6
  configs[2]  B=8 of 1920x1080, DISTS           gains 1.0, 1.3, 1.6
  configs[4]  B=8 of 1920x1080, A-DISTS         gains 1.0, 1.6       (+ B=8 of 256x256)

The reference's result for a pair does not depend on its batch neighbours, so the golden pairs are placed at
chosen slots of a full batch (first, middle, last: in the 2B-image NHWC batch the last slot's y image starts
3.98 GB (f16) / 7.96 GB (f32s) into the activation buffer, beyond 32-bit byte offsets) and the other slots are
filled with device-generated frames.  gain > 1 makes activations grow with depth as ImageNet weights' do
(gain 1.6: relu5_3 mean ~150, max ~1700; gain 1.3: mean ~10, max ~120), which stresses the 16-bit range, the
L2-pool's squaring and the statistics.  Bar: |dscore| <= 1e-4 (BASELINE.json); the measured margins print.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _gold(kind, name, gain):
    tag = "" if gain == 1.0 else "_g%d" % round(gain * 10)
    return np.load(os.path.join(GOLDEN, f"full_{kind}_{name}{tag}.npz"))


def _pairs(g, dev):
    from nerf_qa_amd import synth
    xs, ys = [], []
    for seed, kind in zip(g["seeds"], g["kinds"]):
        x, y = synth.frame_pair(int(seed), int(g["h"]), int(g["w"]), str(kind))
        xs.append(torch.from_numpy(x).to(dev))
        ys.append(torch.from_numpy(y).to(dev))
    return torch.cat(xs), torch.cat(ys)


def _batch_with(gx, gy, slots, b, dev):
    """A batch of b pairs: golden pair i at slots[i], device-generated frames elsewhere."""
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(b, *gx.shape[1:], device=dev, generator=gen)
    y = (x + 0.05 * torch.randn(x.shape, device=dev, generator=gen)).clamp_(0, 1)
    for i, s in enumerate(slots):
        x[s], y[s] = gx[i], gy[i]
    return x, y


def _spec(gain):
    return "synth:1234" if gain == 1.0 else f"synth:1234:{gain:g}"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("gain", [1.0, 1.3, 1.6])
def test_dists_b32_256_vs_reference(gain, dev):
    """configs[1] exactly as stated: the 32 golden pairs ARE the batch."""
    from nerf_qa_amd.DISTS_pytorch import DISTS
    g = _gold("dists", "b32_256", gain)
    x, y = _pairs(g, dev)
    assert x.shape[0] == 32
    # THE GATE: the shipped default ("auto") within 1e-4 of the reference on every pinned weight set.  auto calibrates
    # every rung of its ladder against f32s once per frame-size class with the weights at hand (DISTS_pt.py header;
    # 256x256 frames: 384 pairs of 256x256 / 320x448, three of every eight NeRF-render-like since round 4).  On that
    # content every 16-bit rung shows an outlier of 7.6-8.9e-5 at gain 1.0 (more at 1.3 / 1.6), so all three pinned
    # sets run f32s at this size (round 3, on textures only: f16 / f32m2 / f32s).
    m = DISTS(vgg16_path=_spec(gain)).to(dev).eval()
    assert m.precision == "auto"
    rep = m.calibrate(dev, 256, 256)
    assert rep["size_class"] == 1 and rep["pairs"] == 384
    with torch.no_grad():
        got = m(x, y).cpu().numpy()
    err = np.abs(got - g["score"]).max()
    print(f"\nDISTS B=32 256x256 gain {gain} DEFAULT (auto -> {rep['choice']}; calibration |f16-f32s| max "
          f"{rep['max_abs_diff']:.2e} rms {rep['rms_diff']:.2e}): max|dscore|={err:.2e}")
    assert err <= 1e-4, ("auto", rep, gain, err)
    assert rep["choice"] == "f32s", rep  # (what the pinned sets are known to measure: profiles/r04_cal_classes.txt)
    assert m.precision_for(128, 128, dev) == "f32s"
    assert m.precision_for(100, 120, dev) == "f32s"  # below 128 x 128 pixels: always
    del m
    # per-channel S1 / S2 are quotients with c = 1e-6: on nearly dead channels (variance ~1e-6) a 1e-9 difference in
    # a moment moves S2 by 1e-3, so they get a loose bound; the score (their alpha/beta-weighted sum) is the bar.
    # Explicitly named modes: f32s / f32 to 5e-6.  Forced f16 (which auto admits at gain 1.0 only, and there only from
    # 224x224 pixels up) is held to 1e-4 on this batch at gain 1.0; at gains 1.3 / 1.6 it is OUT OF SPEC by construction (heavy-tailed error: this batch
    # lands at 6e-5 / 1e-4, other seeds at 1.3e-4 / 2e-4, tools/gpu_auto_calibration.py) -- that is why auto does not
    # choose it there; the value is printed, not gated (only a sanity bound).
    for prec, tol, stol in (("f16", 1e-4 if gain == 1.0 else None, None), ("f16w", 1e-4 if gain == 1.0 else None, None), ("f32m4", 1e-4 if gain == 1.0 else None, None), ("f32m", 4e-5, None),
                            ("f32m2", 2e-5, None), ("f32s", 5e-6, 2e-2),
                            ("f32", 5e-6, 2e-2)):
        m = DISTS(precision=prec, vgg16_path=_spec(gain)).to(dev).eval()
        with torch.no_grad():
            got = m(x, y).cpu().numpy()
            s1, s2 = (t.cpu().numpy() for t in m._similarities(x, y))
        err = np.abs(got - g["score"]).max()
        e1, e2 = np.abs(s1 - g["s1"]).max(), np.abs(s2 - g["s2"]).max()
        print(f"\nDISTS B=32 256x256 gain {gain} {prec}: max|dscore|={err:.2e} max|dS1|={e1:.2e} max|dS2|={e2:.2e} "
              f"(scores {g['score'].min():.4f}..{g['score'].max():.4f})"
              + ("" if tol else "   [forced at this gain: out of spec, not chosen by auto]"))
        assert err <= (tol if tol else 5e-4), (prec, gain, err)
        if stol:
            assert e1 <= stol and e2 <= stol, (prec, gain, e1, e2)
        del m


@pytest.mark.parametrize("gain", [1.0, 1.3, 1.6])
def test_dists_1080p_b8_vs_reference(gain, dev):
    """configs[2] at its stated batch: B=8 of 1920x1080, golden pairs at slots 0, 3 and 7."""
    from nerf_qa_amd.DISTS_pytorch import DISTS
    g = _gold("dists", "1080p", gain)
    gx, gy = _pairs(g, dev)
    slots = (0, 3, 7)
    x, y = _batch_with(gx, gy, slots, 8, dev)
    # (gain 1.3, "the ImageNet magnitude", pinned at 1080p since round 4: the default there is f32m; forced f16 / f16w are
    # out of spec at that gain by the calibration's own verdict and only printed)
    loose = gain == 1.3
    for prec, tol in ((None, 1e-4), ("f16", None if loose else 1e-4), ("f16w", None if loose else 1e-4), ("f32m", 4e-5),
                      ("f32s", 2e-5)):  # None = the shipped default (auto)
        m = DISTS(precision=prec, vgg16_path=_spec(gain)).to(dev).eval()
        if prec is None:
            chosen = m.precision_for(1080, 1920, dev)  # class 3: 224 pairs of 720p + 32 of 1080p through every rung
            # (gain 1.3: f32m4's calibration max is 2.0-2.6e-5, which the 1.5e-5 safe-max clause refuses since round 4 -- it
            # reached 6.9e-5 on unseen NeRF-like pairs -- so the verdict is f32m on every box)
            assert chosen == {1.0: "f16", 1.3: "f32m", 1.6: "f32s"}[gain], m.calibrate(dev, 1080, 1920)
            prec = "auto->" + chosen
        with torch.no_grad():
            got = m(x, y)
            s1, s2 = m._similarities(x, y)
        assert got.shape == (8,) and torch.isfinite(got).all()
        sel = got[list(slots)].cpu().numpy()
        err = np.abs(sel - g["score"]).max()
        e1 = np.abs(s1[list(slots)].cpu().numpy() - g["s1"]).max()
        e2 = np.abs(s2[list(slots)].cpu().numpy() - g["s2"]).max()
        print(f"\nDISTS B=8 1080p gain {gain} {prec}: got {sel} ref {g['score']} max|dscore|={err:.2e} "
              f"max|dS1|={e1:.2e} max|dS2|={e2:.2e}")
        assert err <= (tol if tol else 5e-4), (prec, gain, err)
        del m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("gain", [1.0, 1.6])
def test_adists_1080p_b8_vs_reference(gain, dev):
    """configs[4] at its stated batch: A-DISTS, B=8 of 1920x1080 (default f32s), golden pairs at slots 0 and 7."""
    from nerf_qa_amd.ADISTS import ADISTS
    g = _gold("adists", "1080p", gain)
    gx, gy = _pairs(g, dev)
    slots = (0, 7)
    x, y = _batch_with(gx, gy, slots, 8, dev)
    m = ADISTS(vgg16_path=_spec(gain)).to(dev).eval()
    assert m.precision == "auto" and m.precision_for(1080, 1920) == "f32s" and m.precision_for(22, 68) == "f32"
    with torch.no_grad():
        got = m(x, y, as_loss=False)
        again = m(x, y, as_loss=False)
    assert got.shape == (8,) and torch.isfinite(got).all()
    assert torch.equal(got, again)  # no order-dependent accumulation anywhere in the pass
    sel = got[list(slots)].cpu().numpy()
    err = np.abs(sel - g["score"]).max()
    print(f"\nA-DISTS B=8 1080p gain {gain} f32s: got {sel} ref {g['score']} max|dscore|={err:.2e}")
    assert err <= 1e-4, (gain, err)


@pytest.mark.parametrize("gain", [1.0, 1.6])
def test_adists_b8_256_vs_reference(gain, dev):
    from nerf_qa_amd.ADISTS import ADISTS
    g = _gold("adists", "b8_256", gain)
    x, y = _pairs(g, dev)
    m = ADISTS(vgg16_path=_spec(gain)).to(dev).eval()
    with torch.no_grad():
        got = m(x, y, as_loss=False).cpu().numpy()
    err = np.abs(got - g["score"]).max()
    print(f"\nA-DISTS B=8 256x256 gain {gain} f32s: max|dscore|={err:.2e}")
    assert err <= 1e-4, (gain, err)


@pytest.mark.parametrize("gain", [1.0, 1.3, 1.6])
def test_nerf_like_content_vs_reference(gain, dev):
    """NeRF-render-like frames (round 4): constant white / black backgrounds over 40-70 % of the
    frame, smooth gradients, flat frames with floaters -- 12 pairs of 256x256 per weight set, goldens from the imported
    reference (oracle/make_goldens.py `nerf`).  2-4.5 % of their (pair, channel) statistics have EXACTLY zero variance,
    the regime of the 16-bit modes' outliers and of A-DISTS' knife edge, which the full-frame-texture families never
    reach.  The shipped defaults of both metrics must hold 1e-4; every explicitly named DISTS rung is printed."""
    from nerf_qa_amd import synth
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch import DISTS
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import LADDER
    tag = "" if gain == 1.0 else "_g%d" % round(gain * 10)
    g = np.load(os.path.join(GOLDEN, f"nerf_256{tag}.npz"))
    xn, yn = synth.frame_batch([int(s) for s in g["seeds"]], 256, 256, [str(k) for k in g["kinds"]])
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    m = DISTS(vgg16_path=_spec(gain)).to(dev).eval()
    with torch.no_grad():
        got = m(x, y).cpu().numpy()
    chosen = m.precision_for(256, 256, dev)
    err = np.abs(got - g["score"]).max()
    print(f"\nNeRF-like content gain {gain}: DISTS default (auto -> {chosen}) max|dscore|={err:.2e} "
          f"(scores {g['score'].min():.4f}..{g['score'].max():.4f}; dead (pair, channel) fraction per tap {g['dead_frac'].round(3).tolist()})")
    assert err <= 1e-4, (gain, chosen, err)
    del m
    for prec in LADDER + ("f32",):
        mm = DISTS(precision=prec, vgg16_path=_spec(gain)).to(dev).eval()
        with torch.no_grad():
            e = np.abs(mm(x, y).cpu().numpy() - g["score"])
        print(f"   {prec:6s}: max|dscore|={e.max():.2e} at {g['kinds'][int(e.argmax())]}")
        if prec in ("f32s", "f32"):  # (3.6e-6 on the flat-plus-floaters pairs in BOTH: exactly dead channels, summation order)
            assert e.max() <= 1e-5, (prec, gain, e.max())
        del mm
    a = ADISTS(vgg16_path=_spec(gain)).to(dev).eval()
    with torch.no_grad():
        agot = a(x, y, as_loss=False).cpu().numpy()
    aerr = np.abs(agot - g["adists"])
    print(f"   A-DISTS default (auto -> {a.precision_for(256, 256)}): max|dscore|={aerr.max():.2e} at {g['kinds'][int(aerr.argmax())]}")
    assert aerr.max() <= 1e-4, (gain, aerr)


@pytest.mark.parametrize("h,w,prec", [(1080, 1920, "f32s"), (1080, 1920, "f16"), (1080, 1920, "f32m"), (2160, 3840, "f16")],
                         ids=["1080p_f32s", "1080p_f16", "1080p_f32m", "4k_f16"])
def test_forward_once_full_size(h, w, prec, dev):
    """forward_once (DISTS_pt.py:91-103) at sizes whose float taps exceed 2^31 bytes per image: the NCHW export
    has no 32-bit offset limit; values equal the pyramid's own NHWC taps."""
    from nerf_qa_amd import ops
    from nerf_qa_amd.DISTS_pytorch import DISTS
    m = DISTS(precision=prec, vgg16_path="synth:1234").to(dev).eval()
    x = torch.rand(1, 3, h, w, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    with torch.no_grad():
        feats = m.forward_once(x)
        taps = ops.vgg_pyramid(x, m._packed_weights(dev, prec), prec)
    assert feats[0] is x and [f.shape[1] for f in feats] == [3, 64, 128, 256, 512, 512]
    assert [t.dtype for t in taps] == [torch.float32 if ops.tap_prec(prec, k) in (0, 3) else torch.float16 for k in range(5)]
    for f, t in zip(feats[1:], taps):
        assert f.dtype == torch.float32 and f.shape == (1, t.shape[3], t.shape[1], t.shape[2])
        assert torch.equal(f, t.permute(0, 3, 1, 2).float())
        del f
    del feats, taps
    torch.cuda.empty_cache()


def test_synthetic_video_frames_are_index_addressed(dev):
    """configs[3]'s frames: frame i is a pure function of its index, whatever batch it is generated in."""
    from nerf_qa_amd import video
    a_ref, a_ren = video.synthetic_frames(range(96, 104), 270, 480, dev)
    b_ref, b_ren = video.synthetic_frames([100], 270, 480, dev)
    assert torch.equal(a_ref[4], b_ref[0]) and torch.equal(a_ren[4], b_ren[0])
    assert not torch.equal(a_ref[4], a_ref[5])
    assert 0.0 <= a_ren.min().item() and a_ren.max().item() <= 1.0 and abs(a_ref.mean().item() - 0.5) < 1e-2


def test_rccl_one_rank():
    """The collective calls bench.py / sharding.py make at N > 1, on RCCL itself with one rank (a 1-GPU box cannot
    host two): init with device_id, barrier, all_gather into the per-rank timing list, all_gather_into_tensor, the
    object broadcast of DISTS' calibration verdict."""
    import subprocess
    import sys
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from nerf_qa_amd import sharding\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29537', RANK='0', WORLD_SIZE='1')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "dist.barrier()\n"
        "t = torch.tensor([1.5], dtype=torch.float64, device=dev)\n"
        "al = [torch.zeros_like(t)]; dist.all_gather(al, t)\n"
        "scores = torch.arange(10, dtype=torch.float32, device=dev)\n"
        "out = sharding.gather_scores(scores, 10)\n"
        "full = sharding.score_frames_sharded(lambda lo, hi: scores[lo:hi] * 2, 10, 4, dev)\n"
        "assert torch.equal(out, scores) and al[0].item() == 1.5 and torch.equal(full, scores * 2)\n"
        # the calibration verdict's object broadcast (sharding.agree_precision), on a real `auto` module
        "os.environ['NQA_CAL_CACHE'] = 'off'\n"
        "from nerf_qa_amd.DISTS_pytorch import DISTS\n"
        "net = DISTS(vgg16_path='synth:1234').to(dev).eval()\n"
        "mode = sharding.agree_precision(net, 160, 192, dev)\n"
        "assert mode == net.precision_for(160, 192, dev) and net._agreed_report['agreed_over_ranks'] == 1\n"
        "dist.destroy_process_group(); print('RCCL one-rank ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL one-rank ok" in r.stdout, r.stdout + r.stderr


def test_former_knife_edge_pairs(dev):
    """The four pairs on which round 1's f32s A-DISTS sat 4e-4 from the oracle (tools/gpu_stress.py).  The imported
    reference returns the same value for them in f32 with 8 threads / 1 thread / oneDNN off / channels_last and in
    float64 (oracle/knife_edge_study.py; spread <= 2.3e-7), so that value is THE answer and the HIP default must match.
    The default is "auto": frames this small run with exact-f32 products, so the match no longer hangs on the rounding
    pattern of one kernel (round 2: an equally accurate conv1_1 of another summation order flipped the 22x68 pair in
    f32s); f32s on them is recorded beside it."""
    from nerf_qa_amd import synth
    from nerf_qa_amd.ADISTS import ADISTS
    g = np.load(os.path.join(GOLDEN, "knife_edge_adists.npz"))
    assert (g["reference"].max(1) - g["reference"].min(1)).max() <= 1e-6
    m = ADISTS(vgg16_path="synth:1234").to(dev).eval()
    ms = ADISTS(vgg16_path="synth:1234", precision="f32s").to(dev).eval()
    for h, w, seed, kind, ref in zip(g["h"], g["w"], g["seed"], g["kind"], g["reference"]):
        x, y = synth.frame_pair(int(seed), int(h), int(w), str(kind))
        assert m.precision_for(int(h), int(w)) == "f32"
        with torch.no_grad():
            got = m(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), as_loss=False).item()
            got_s = ms(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), as_loss=False).item()
        print(f"\n{h}x{w} seed {seed}: hip auto(f32) {got:.7f} f32s {got_s:.7f} reference {ref[0]:.7f} (f64 {ref[-1]:.7f})")
        assert abs(got - float(ref[0])) <= 2e-5, (h, w, seed, got, ref)
