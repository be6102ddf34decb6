"""The nn.Module boundary (drop-in for nerf_qa.DISTS_pytorch.DISTS) on the GPU."""
import io
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from nerf_qa_amd.DISTS_pytorch import DISTS
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return DISTS().to(dev).eval()


def _pair(h, w, seeds=(0, 1, 2)):
    from nerf_qa_amd import synth
    x, y = synth.frame_batch(list(seeds), h, w)
    return torch.from_numpy(x), torch.from_numpy(y)


def test_forward_matches_oracle(model, oracle_convs, dev):
    from oracle import dists_oracle
    x, y = _pair(80, 112)
    ref = dists_oracle.dists(x, y, oracle_convs, model.alpha.detach().cpu(), model.beta.detach().cpu())
    with torch.no_grad():
        got = model(x.to(dev), y.to(dev))
    assert got.shape == (3,) and got.dtype == torch.float32
    assert (got.cpu() - ref).abs().max().item() <= 1e-4
    with torch.no_grad():
        avg = model(x.to(dev), y.to(dev), batch_average=True)
    assert avg.dim() == 0 and abs(avg.item() - ref.mean().item()) <= 1e-4
    # warp / certainty are accepted and ignored, as in DISTS_pt.py:105
    with torch.no_grad():
        again = model(x.to(dev), y.to(dev), warp=object(), certainty=object())
    assert torch.equal(again, got)


def test_forward_once_and_from_feats(model, oracle_convs, dev):
    from oracle import dists_oracle
    x, y = _pair(64, 64, seeds=(5,))
    with torch.no_grad():
        f0, f1 = model.forward_once(x.to(dev)), model.forward_once(y.to(dev))
        a = model.forward_from_feats(f0, f1)
        b = model(x.to(dev), y.to(dev))
    assert [tuple(f.shape) for f in f0] == [(1, 3, 64, 64), (1, 64, 64, 64), (1, 128, 32, 32), (1, 256, 16, 16),
                                            (1, 512, 8, 8), (1, 512, 4, 4)]
    assert f0[0].data_ptr() == x.to(dev).data_ptr() or torch.equal(f0[0].cpu(), x)
    assert (a - b).abs().max().item() < 2e-6
    ref = dists_oracle.vgg_pyramid(x, oracle_convs)
    for got, r in zip(f0[1:], ref[1:]):
        assert (got.cpu() - r).abs().max().item() <= 4e-3 * r.abs().max().item()


def test_alpha_beta_gradients(model, dev):
    """The fine-tuning loop (run_nerf_qa.py:433-461) needs d score / d alpha,beta."""
    x, y = _pair(48, 48, seeds=(7, 8))
    model.zero_grad()
    score = model(x.to(dev), y.to(dev))
    assert score.requires_grad
    score.sum().backward()
    assert model.alpha.grad is not None and model.beta.grad is not None
    with torch.no_grad():
        fused = model(x.to(dev), y.to(dev))
    assert (fused - score.detach()).abs().max().item() < 2e-6
    # finite-difference check on one alpha entry
    s1, s2 = model._similarities(x.to(dev), y.to(dev))
    a = model.alpha.detach().view(-1).double()
    b_ = model.beta.detach().view(-1).double()
    w = a.sum() + b_.sum()
    j = 10
    d = (-(s1[:, j].double()) / w + ((a * s1.double()).sum(1) + (b_ * s2.double()).sum(1)) / w ** 2).sum()
    assert abs(model.alpha.grad.view(-1)[j].item() - d.item()) < 1e-4 * max(1.0, abs(d.item()))
    model.zero_grad()


def test_project_weights_and_pickle(model, dev):
    import copy
    m = copy.deepcopy(model)
    m.project_weights()
    assert abs((m.alpha.sum() + m.beta.sum()).item() - 1) < 1e-5
    assert m.alpha[:, :3].min().item() >= 0.019
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    x, y = _pair(32, 40, seeds=(2,))
    with torch.no_grad():
        assert torch.equal(m2(x.to(dev), y.to(dev)), m(x.to(dev), y.to(dev)))


def test_refuses_cpu_and_require_grad(model, dev):
    from nerf_qa_amd import NqaError
    x, y = _pair(32, 32, seeds=(1,))
    with pytest.raises(NqaError):
        model(x, y)
    # require_grad=True on inputs that carry no gradient: the value, as the reference returns it (DISTS_pt.py:106-108);
    # the gradient path itself is tests/test_gpu_backward.py
    with torch.no_grad():
        assert torch.equal(model(x.to(dev), y.to(dev), require_grad=True), model(x.to(dev), y.to(dev)))
    with pytest.raises(ValueError):
        model(x.to(dev), y[:, :, :16].to(dev))


def test_original_and_softmax_variants(model, oracle_convs, dev):
    """DISTS_pt_original (0-d for B=1, clamped alpha/beta) and DISTS_pt_softmax agree with the
    canonical score when their weight transforms are identities."""
    import pandas as pd  # noqa: F401
    from nerf_qa_amd import config as cfgmod
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_original import DISTS as DO
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_softmax import DISTS as DS
    cfg = cfgmod.config()
    cfg.weight_lower_bound, cfg.alpha_beta_ratio, cfg.dists_weight_norm, cfg.detach_beta = 0.0, 1.0, "off", "False"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mo, ms = DO().to(dev).eval(), DS().to(dev).eval()
    x, y = _pair(64, 64, seeds=(11, 12))
    x, y = x.to(dev), y.to(dev)
    with torch.no_grad():
        ref = model(x, y)
        a = mo(x, y)
        b = ms(x, y)
        one = mo(x[:1], y[:1])
    assert a.shape == (2,) and one.dim() == 0
    assert (a - ref).abs().max().item() < 2e-6
    assert (b - ref).abs().max().item() < 2e-5          # softmax(log(w + 1e-10)) reproduces w to ~1e-7
    assert mo.original_alpha.device == x.device
    # options: relu + detach_beta leave a graph to alpha only
    cfg.dists_weight_norm, cfg.detach_beta = "relu+w_sum_detach", "True"
    s = mo(x, y).sum()
    mo.zero_grad()
    s.backward()
    assert mo.alpha.grad is not None and (mo.beta.grad is None or float(mo.beta.grad.abs().sum()) == 0.0)
    cfg.dists_weight_norm, cfg.detach_beta = "off", "False"


def test_nerfqa_model_head(dev):
    import pandas as pd
    from nerf_qa_amd import config as cfgmod
    from nerf_qa_amd.model_stats import NeRFQAModel
    cfg = cfgmod.config()
    rng = np.random.default_rng(0)
    d = rng.uniform(0.05, 0.4, 40)
    df = pd.DataFrame({"DISTS": d, "MOS": 5 - 8 * d + 0.01 * rng.standard_normal(40)})
    x, y = _pair(48, 48, seeds=(1, 2))
    for kind in ("linear", "sqrt", "logistic"):
        cfg.regression_type = kind
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = NeRFQAModel(df).to(dev)
        scores, dists_scores = m(x.to(dev), y.to(dev))
        assert scores.shape == dists_scores.shape == (2,)
        assert torch.isfinite(scores).all()
        (scores.sum() + 1e-3 * m.entropy_loss()).backward()
        assert m.dists_model.alpha.grad is not None
        if kind == "linear":
            assert abs(m.dists_weight.item() + 8) < 0.5 and abs(m.dists_bias.item() - 5) < 0.2
    cfg.regression_type = "linear"


def test_oversized_batches_are_sliced(monkeypatch):
    """A batch whose scratch would exceed NQA_MAX_WORKSPACE_GB runs in slices with identical results."""
    import warnings
    from nerf_qa_amd import ops, synth
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch import DISTS
    dev = torch.device("cuda:0")
    xn, yn = synth.frame_batch(list(range(7)), 64, 80)
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m, a = DISTS().to(dev).eval(), ADISTS().to(dev).eval()
    with torch.no_grad():
        whole, awhole = m(x, y), a(x, y, as_loss=False)
        per_pair = ops.lib().nqa_adists_workspace_bytes(1, 64, 80, 3)
        monkeypatch.setenv("NQA_MAX_WORKSPACE_GB", str(2.5 * per_pair / (1 << 30)))  # room for two pairs at a time
        assert ops._max_pairs(lambda n: ops.lib().nqa_adists_workspace_bytes(n, 64, 80, 3), 7) == 2  # 2,2,2,1
        sliced, asliced = m(x, y), a(x, y, as_loss=False)
        amap = a(x, y, as_map=True)
    assert torch.equal(sliced, whole) and torch.equal(asliced, awhole)
    assert amap.shape == (7, 7, 64, 80)
