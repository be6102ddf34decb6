"""The training-time variants and the NeRFQAModel head against goldens frozen from the imported reference
(oracle/make_goldens.py: DISTS_pt_original, DISTS_pt_softmax, model_stats.NeRFQAModel under several run configs)."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN

G = np.load(os.path.join(GOLDEN, "variants_64x64.npz"))
KINDS = ("linear", "sqrt", "logistic")


def _cfg(lb, ratio, norm, det, kind="linear"):
    from nerf_qa_amd import config as cfgmod
    c = cfgmod.config()
    c.weight_lower_bound, c.alpha_beta_ratio, c.dists_weight_norm, c.detach_beta = float(lb), float(ratio), str(norm), str(det)
    c.subjective_score_type, c.regression_type = "MOS", kind
    return c


def _df():
    import pandas as pd
    return pd.DataFrame({"DISTS": G["train_dists"], "MOS": G["train_mos"]})


@pytest.mark.parametrize("kind", KINDS)
def test_head_fit_matches_reference(kind):
    """model_stats.py:27-61: the regression that initialises the head (no GPU involved)."""
    from nerf_qa_amd.model_stats import NeRFQAModel
    _cfg(1e-4, 1.0, "relu", "False", kind)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = NeRFQAModel(_df())
    got = [m.b1, m.b2, m.b3, m.b4] if kind == "logistic" else [m.dists_weight, m.dists_bias]
    got = np.array([p.item() for p in got])
    assert np.allclose(got, G[f"head_{kind}_params"], rtol=1e-5, atol=1e-6), (got, G[f"head_{kind}_params"])
    _cfg(0.0, 1.0, "off", "False")


@pytest.mark.gpu
def test_variants_and_head_match_reference(monkeypatch):
    from nerf_qa_amd import synth
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_original import DISTS as DO
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_softmax import DISTS as DS
    from nerf_qa_amd.model_stats import NeRFQAModel
    monkeypatch.setenv("NQA_PRECISION", "f32")  # exact-f32 convolutions: the comparison is about the head arithmetic
    dev = torch.device("cuda:0")
    xn, yn = synth.frame_batch([int(s) for s in G["seeds"]], int(G["h"]), int(G["w"]))
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    tol = 5e-6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, ((lb, ratio), norm, det) in enumerate(zip(G["configs"], G["norms"], G["detach"])):
            _cfg(lb, ratio, norm, det)
            m = DO().to(dev).eval()
            with torch.no_grad():
                s, one = m(x, y), m(x[:1], y[:1])
            assert one.dim() == 0 and np.abs(s.cpu().numpy() - G[f"orig{i}_score"]).max() <= tol
            assert abs(one.item() - float(G[f"orig{i}_one"])) <= tol
            m.project_weights()
            assert np.abs(m.alpha.data.cpu().numpy().reshape(-1) - G[f"orig{i}_alpha"]).max() <= 1e-8
            assert np.abs(m.beta.data.cpu().numpy().reshape(-1) - G[f"orig{i}_beta"]).max() <= 1e-8
            with torch.no_grad():
                assert np.abs(m(x, y).cpu().numpy() - G[f"orig{i}_projected"]).max() <= tol
        _cfg(0.0, 1.0, "softmax", "False")
        with torch.no_grad():
            assert np.abs(DS().to(dev).eval()(x, y).cpu().numpy() - G["soft_score"]).max() <= tol
        for kind in KINDS:
            _cfg(1e-4, 1.0, "relu", "False", kind)
            M = NeRFQAModel(_df()).to(dev).eval()
            with torch.no_grad():
                scores, ds = M(x, y)
                ent = M.entropy_loss()
            assert np.abs(ds.cpu().numpy() - G[f"head_{kind}_dists"]).max() <= tol
            assert np.abs(scores.cpu().numpy() - G[f"head_{kind}_scores"]).max() <= 2e-4  # the head multiplies by ~8
            assert abs(ent.item() - float(G[f"head_{kind}_entropy"])) <= 1e-4 * abs(float(G[f"head_{kind}_entropy"]))
    _cfg(0.0, 1.0, "off", "False")


@pytest.mark.gpu
def test_mode_model_matches_reference(monkeypatch):
    """nerf_qa/model.py:22-56 (the wandb.config.mode variant four entry scripts import): scores of every mode."""
    from nerf_qa_amd import config as cfgmod, synth
    from nerf_qa_amd.model import NeRFQAModel
    monkeypatch.setenv("NQA_PRECISION", "f32")
    dev = torch.device("cuda:0")
    xn, yn = synth.frame_batch([int(s) for s in G["seeds"]], int(G["h"]), int(G["w"]))
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    c = _cfg(1e-4, 1.0, "relu", "False")
    try:
        for mode in ("linear", "sqrt", "softmax", "softmax+sqrt"):
            c.mode = mode
            M = NeRFQAModel(_df()).to(dev).eval()
            with torch.no_grad():
                scores, ds = M(x, y)
            key = "mode_" + mode.replace("+", "_")
            assert np.abs(ds.cpu().numpy() - G[key + "_dists"]).max() <= 5e-6
            assert np.abs(scores.cpu().numpy() - G[key + "_scores"]).max() <= 2e-4  # the head multiplies by ~8
    finally:
        c.mode = "linear"
        _cfg(0.0, 1.0, "off", "False")
