"""nerf_qa_amd.install_alias(): the reference's import lines and whole-module pickles work UNMODIFIED
(run_nerf_qa.py:29,33,502; reeval.py:83; nerf_qa/data.py:34; prep.py:27-28).  CPU only, no kernel runs."""
import importlib
import io
import sys

import numpy as np
import pandas as pd
import pytest
import torch

import nerf_qa_amd
from test_boundary import REFERENCE_IMPORTS


@pytest.fixture()
def alias():
    assert "nerf_qa" not in sys.modules or not hasattr(sys.modules["nerf_qa"], "__file__")
    mods = nerf_qa_amd.install_alias()
    yield mods
    nerf_qa_amd.remove_alias()
    assert "nerf_qa.model_stats" not in sys.modules


@pytest.mark.parametrize("mod,names,who", REFERENCE_IMPORTS, ids=[m for m, _, _ in REFERENCE_IMPORTS])
def test_reference_import_lines_resolve_unmodified(alias, mod, names, who):
    m = importlib.import_module(mod)  # the reference's own module path
    assert m.__name__.startswith("nerf_qa_amd"), (mod, m.__name__)
    ns = {}
    exec(f"from {mod} import {', '.join(names)}", ns)  # the line as the reference's scripts have it
    for n in names:
        assert ns[n] is getattr(m, n)


def test_the_scripts_import_block(alias):
    ns = {}
    exec("from nerf_qa.DISTS_pytorch.DISTS_pt_original import DISTS, prepare_image\n"   # run_nerf_qa.py:29
         "from nerf_qa.model_stats import NeRFQAModel\n"                                # run_nerf_qa.py:33
         "from nerf_qa.ADISTS import ADISTS\n"                                          # prep.py:28
         "import nerf_qa.DISTS_pytorch.DISTS_pt as dp\n", ns)
    assert ns["NeRFQAModel"].__module__ == "nerf_qa_amd.model_stats" and ns["dp"].DISTS.__module__.endswith("DISTS_pt")


def _train_df():
    rng = np.random.default_rng(0)
    d = rng.uniform(0.05, 0.4, 40)
    return pd.DataFrame({"DISTS": d, "MOS": 5 - 8 * d + rng.normal(0, 0.05, 40)})


def _as_reference_pickle(obj):
    """torch.save(whole module) in the legacy (plain pickle, protocol 2) format, with every class path renamed to the
    reference's module names -- the bytes a script of the reference would have written (protocol-2 GLOBAL opcodes are
    text lines, so the rename is a byte substitution)."""
    buf = io.BytesIO()
    torch.save(obj, buf, _use_new_zipfile_serialization=False)
    raw = buf.getvalue()
    assert b"cnerf_qa_amd.model_stats\nNeRFQAModel\n" in raw or b"cnerf_qa_amd." in raw
    return raw.replace(b"cnerf_qa_amd.", b"cnerf_qa.")


def test_whole_module_pickle_round_trip(alias):
    from nerf_qa.model_stats import NeRFQAModel  # noqa: the alias
    model = NeRFQAModel(_train_df(), vgg16_path="synth:1234")
    raw = _as_reference_pickle(model)
    assert b"cnerf_qa.model_stats\nNeRFQAModel" in raw and b"nerf_qa_amd.model_stats" not in raw
    back = torch.load(io.BytesIO(raw), weights_only=False)  # reeval.py:83
    assert type(back) is NeRFQAModel and type(back.dists_model).__module__ == "nerf_qa_amd.DISTS_pytorch.DISTS_pt_original"
    assert torch.equal(back.dists_weight, model.dists_weight) and torch.equal(back.dists_model.alpha, model.dists_model.alpha)
    w0 = [m.weight for m in model.dists_model._conv_modules()]
    w1 = [m.weight for m in back.dists_model._conv_modules()]
    assert len(w1) == 13 and all(torch.equal(a, b) for a, b in zip(w0, w1))
    # device scratch never travels; the private fields are rebuilt
    assert back.dists_model._packed == {} and back.dists_model.precision == model.dists_model.precision


def test_pickle_leaves_every_device_cache_behind():
    """after a require_grad=True step the module carries ~60 MB of packed CUDA blobs (`_bwd_blobs`), and since
    round 4 calibration deltas / a process-group agreement / memo fields; none of them may enter torch.save(model)
    (run_nerf_qa.py:502) -- a CPU-only torch.load (reeval.py:83) could not even open the CUDA ones."""
    from nerf_qa_amd.DISTS_pytorch import DISTS
    m = DISTS(vgg16_path="synth:1234")
    m.__dict__["_bwd_blobs"] = (("key",), {1: torch.zeros(4)}, torch.zeros(4))  # (what autograd._backward_blobs leaves)
    m._deltas = {0: {"f16": (torch.zeros(2, 1475), torch.zeros(2, 1475))}}
    m._agreed = {3: (("key",), "f32m")}
    m._live_weights(torch.device("cpu"))
    m._vgg_digest()
    state = m.__getstate__()
    assert "_bwd_blobs" not in state and "_live" not in state and "_digest" not in state and "_packed" not in state
    assert state["_deltas"] == {} and state["_agreed"] == {} and state["_auto"] is None
    buf = io.BytesIO()
    torch.save(m, buf)
    assert len(buf.getvalue()) < 60e6  # the 58.9 MB of Conv2d weights and little else
    back = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
    assert back._deltas == {} and back._agreed == {} and not hasattr(back, "_bwd_blobs")


def test_module_pickled_without_this_builds_private_fields(alias):
    """A module object written by the reference's OWN class has none of this build's private attributes
    (precision, _packed, _ws, vgg_source): __setstate__ fills them in."""
    from nerf_qa.ADISTS import ADISTS
    from nerf_qa.DISTS_pytorch.DISTS_pt import DISTS
    for cls, priv in ((DISTS, ("precision", "vgg_source", "_packed", "_ws")),
                      (ADISTS, ("precision", "vgg_source", "_packed", "_ws"))):
        m = cls(vgg16_path="synth:1234")
        state = {k: v for k, v in m.__dict__.items() if k not in priv}
        back = cls.__new__(cls)
        back.__setstate__(state)
        for k in priv:
            assert hasattr(back, k), (cls, k)
        assert len(back._conv_modules()) == 13 and back.precision == "auto"


def test_without_alias_the_reference_names_do_not_resolve():
    nerf_qa_amd.remove_alias()
    if "nerf_qa" in sys.modules:
        pytest.skip("a real nerf_qa package is importable here")
    with pytest.raises(ImportError):
        importlib.import_module("nerf_qa.model_stats")
