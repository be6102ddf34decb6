"""End-to-end parity of the HIP DISTS path against the oracle and the golden vectors.

The bar (BASELINE.json north_star): |score_hip - score_ref| <= 1e-4 on identical fp32
frame pairs.  Because the stand-in weights compress the score range, S1/S2 are gated too.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
# north_star tolerance.  f32 (exact-f32 MFMA) and f16 (the default) are held to it.  bf16 is
# an opt-in mode measured at up to 3.1e-4 on small frames with the stand-in weights (its
# 8-bit mantissa rounds every stored activation at 2^-9); it is held to 5e-4 and
# documented in DESIGN.md as NOT meeting the bar.
SCORE_TOL = {"f32": 1e-4, "f32s": 1e-4, "f32m": 1e-4, "f32m2": 1e-4, "f32m4": 1e-4, "f16w": 1e-4, "f16": 1e-4, "bf16": 5e-4}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def packed(np_convs, dev):
    from nerf_qa_amd import ops
    return {p: ops.pack_vgg_weights(np_convs, p).to(dev) for p in ("f32", "f32s", "f32m", "f32m2", "f32m4", "f16w", "f16", "bf16")}


def _load_case(path):
    from nerf_qa_amd import synth
    g = np.load(path)
    x, y = synth.frame_batch([int(s) for s in g["seeds"]], int(g["h"]), int(g["w"]), [str(k) for k in g["kinds"]])
    return g, torch.from_numpy(x), torch.from_numpy(y)


DISTS_GOLD = sorted(glob.glob(os.path.join(GOLDEN, "dists_*.npz")))


# s_tol gates the per-channel maximum for f32 and the channel-mean for the 16-bit paths: at
# stage 5 of a 20x20 input the statistics run over 4 pixels, where a single rounded
# activation moves one channel's S2 by O(0.1) while the score stays within 1e-5.
@pytest.mark.parametrize("prec,s_tol", [("f32", 5e-4), ("f32s", 5e-4), ("f32m", 5e-3), ("f32m2", 5e-3), ("f32m4", 5e-3), ("f16w", 5e-3), ("f16", 5e-3), ("bf16", 4e-2)])
@pytest.mark.parametrize("path", DISTS_GOLD, ids=[os.path.basename(p)[:-4] for p in DISTS_GOLD])
def test_dists_vs_golden(path, prec, s_tol, packed, alpha_beta, dev):
    from nerf_qa_amd import ops
    g, x, y = _load_case(path)
    s1, s2 = ops.dists_forward(x.to(dev), y.to(dev), packed[prec], prec)
    alpha, beta = alpha_beta
    score = ops.dists_score(s1, s2, alpha.to(dev), beta.to(dev)).cpu().numpy()
    d = np.abs(score - g["score"]).max()
    red = np.max if prec in ("f32", "f32s") else np.mean
    e1 = red(np.abs(s1.cpu().numpy() - g["s1"]))
    e2 = red(np.abs(s2.cpu().numpy() - g["s2"]))
    print(f"\n{os.path.basename(path)} [{prec}] |dscore|={d:.2e} |dS1|={e1:.2e} |dS2|={e2:.2e}")
    assert d <= SCORE_TOL[prec], f"|dscore| {d:.3e} > {SCORE_TOL[prec]}"
    assert e1 <= s_tol and e2 <= s_tol


@pytest.mark.parametrize("prec,rtol", [("f32", 3e-5), ("f32s", 3e-5), ("f32m", 2e-3), ("f32m2", 2e-3), ("f32m4", 2e-3), ("f16w", 2e-3), ("f16", 4e-3), ("bf16", 3e-2)])
def test_pyramid_taps(prec, rtol, packed, oracle_convs, dev):
    """forward_once: every tapped map against the oracle, odd size so every stage is ragged."""
    from nerf_qa_amd import ops, synth
    from oracle import dists_oracle
    x, _ = synth.frame_batch([5, 6], 97, 131)
    x = torch.from_numpy(x)
    ref = dists_oracle.vgg_pyramid(x, oracle_convs)[1:]
    taps = ops.vgg_pyramid(x.to(dev), packed[prec], prec)
    for k, (t, r) in enumerate(zip(taps, ref)):
        got = ops.nhwc_to_nchw_f32(t, ops.tap_prec(prec, k)).cpu()  # ("f32m": half taps 1..3, float taps 4..5)
        assert got.shape == r.shape
        err = (got - r).abs().max().item() / r.abs().max().item()
        print(f"\n tap {k + 1} [{prec}] rel err {err:.2e}")
        assert err <= rtol


def test_identical_inputs_score_zero(packed, alpha_beta, dev):
    from nerf_qa_amd import ops, synth
    x, _ = synth.frame_batch([9], 64, 80)
    x = torch.from_numpy(x).to(dev)
    for prec in ("f32", "f32s", "f32m", "f32m2", "f32m4", "f16w", "f16", "bf16"):
        s1, s2 = ops.dists_forward(x, x.clone(), packed[prec], prec)
        alpha, beta = alpha_beta
        score = ops.dists_score(s1, s2, alpha.to(dev), beta.to(dev))
        assert abs(score.item()) < 2e-6


def test_symmetry_and_batch_independence(packed, alpha_beta, dev):
    """score(x,y)==score(y,x); a pair's score does not depend on its batch neighbours."""
    from nerf_qa_amd import ops, synth
    x, y = synth.frame_batch([1, 2, 3], 48, 64)
    x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    alpha, beta = [t.to(dev) for t in alpha_beta]
    for prec in ("f32", "bf16"):
        a = ops.dists_score(*ops.dists_forward(x, y, packed[prec], prec), alpha, beta)
        b = ops.dists_score(*ops.dists_forward(y, x, packed[prec], prec), alpha, beta)
        c = ops.dists_score(*ops.dists_forward(x[1:2], y[1:2], packed[prec], prec), alpha, beta)
        assert (a - b).abs().max().item() < 1e-6
        assert abs(a[1].item() - c[0].item()) < 1e-6
