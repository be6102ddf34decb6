"""The drop-in boundary on CPU: every symbol the reference's scripts import from the hot-path modules exists here,
and the host-side parts (prepare_image, project_weights, head fits, video columns) match goldens frozen from
the imported reference (oracle/make_goldens.py).  No kernel runs in this file."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

# Frozen from /root/reference (grep "from nerf_qa.(DISTS_pytorch|ADISTS|model|model_stats) ... import X" over the
# entry scripts and nerf_qa/data*.py): (reference module, names, who imports them)
REFERENCE_IMPORTS = [
    ("nerf_qa.DISTS_pytorch.DISTS_pt", ("DISTS", "prepare_image"),
     "data_prep.py:30 nerf_qa/data.py:34 nerf_qa/nerf_nr_qa_prep*.py prep.py:27 run.py:22 run_test2*.py train-nr.py:27"),
    ("nerf_qa.DISTS_pytorch.DISTS_pt_original", ("DISTS", "prepare_image"),
     "nerf_qa/data_fr.py:34 run_nerf_qa.py:29 test2_prep.py:32 nerf_qa/model.py:43 nerf_qa/model_stats.py:66"),
    ("nerf_qa.DISTS_pytorch.DISTS_pt_softmax", ("DISTS",), "nerf_qa/model.py:41 nerf_qa/model_stats.py:64"),
    ("nerf_qa.DISTS_pytorch", ("DISTS",), "nerf_qa/DISTS_pytorch/__init__.py:1"),
    ("nerf_qa.ADISTS", ("ADISTS",), "prep.py:28 test2_prep.py:33 nerf_qa/nerf_nr_qa_prep.py:9 nerf_nr_qa_prep_4.py:10"),
    ("nerf_qa.ADISTS.ADISTS", ("ADISTS",), "nerf_qa/ADISTS/__init__.py:1"),
    ("nerf_qa.model", ("NeRFQAModel",), "run.py:26 run_test2.py:27 run_test2_cross.py:29 run_test2_sf.py:29"),
    ("nerf_qa.model_stats", ("NeRFQAModel",), "reeval.py:32 run_final.py:32 run_nerf_qa.py:33 run_test2_stats.py:29"),
]


@pytest.mark.parametrize("mod,names,who", REFERENCE_IMPORTS, ids=[m for m, _, _ in REFERENCE_IMPORTS])
def test_every_reference_import_resolves(mod, names, who):
    ours = importlib.import_module(mod.replace("nerf_qa", "nerf_qa_amd", 1))
    for name in names:
        assert hasattr(ours, name), f"{mod}.{name} (imported by {who}) is missing from the drop-in"


def test_reference_signatures():
    """Argument names and defaults of the callable surface."""
    import inspect
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import DISTS, prepare_image
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_original import DISTS as DO, prepare_image as prepare_image_orig

    def params(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values() if p.name != "self"]
    E = inspect.Parameter.empty
    assert params(DISTS.forward) == [("x", E), ("y", E), ("require_grad", False), ("batch_average", False),
                                     ("warp", None), ("certainty", None)]          # DISTS_pt.py:105
    assert params(DISTS.forward_from_feats) == [("feats0", E), ("feats1", E), ("batch_average", False)]  # :181
    assert params(DISTS.__init__)[:2] == [("load_weights", True), ("from_feats", False)]                 # :28
    assert params(DO.forward) == [("x", E), ("y", E), ("require_grad", False), ("batch_average", False)]  # _original:97
    assert params(ADISTS.forward) == [("x", E), ("y", E), ("as_loss", True), ("as_map", False)]           # ADISTS.py:137
    assert params(ADISTS.__init__)[:1] == [("window_size", 21)]
    assert params(prepare_image) == [("image", E), ("resize", True), ("keep_aspect_ratio", False)]        # :210
    assert params(prepare_image_orig) == [("image", E), ("resize", True)]                                 # _original:140


def test_prepare_image_matches_reference():
    """Both prepare_image variants against the reference's, run on the same seeded PIL images."""
    from PIL import Image
    from nerf_qa_amd import synth
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import prepare_image
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_original import prepare_image as prepare_image_orig
    g = np.load(os.path.join(GOLDEN, "prepare_image.npz"))
    for k in range(5):
        h, w = (int(v) for v in g[f"p{k}_hw"])
        arr = (synth.uniform(900 + k, h * w * 3).reshape(h, w, 3) * 256).astype(np.uint8)
        img = Image.fromarray(arr, "RGB")
        calls = {"sq": prepare_image(img), "keep": prepare_image(img, resize=True, keep_aspect_ratio=True),
                 "none": prepare_image(img, resize=False), "orig": prepare_image_orig(img),
                 "orig_none": prepare_image_orig(img, resize=False)}
        for tag, t in calls.items():
            assert t.dtype == torch.float32 and list(t.shape) == g[f"p{k}_{tag}_shape"].tolist(), (k, tag, t.shape)
            assert abs(t.double().sum().item() - float(g[f"p{k}_{tag}_sum"])) <= 1e-6 * float(g[f"p{k}_{tag}_sum"])
            assert np.array_equal((t[0, :, ::37, ::41] * 255).round().to(torch.uint8).numpy(), g[f"p{k}_{tag}_u8"])


def test_canonical_project_weights_matches_reference():
    """DISTS.project_weights (DISTS_pt.py:82-89) on the published and on perturbed alpha/beta."""
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import DISTS
    g = np.load(os.path.join(GOLDEN, "variants_64x64.npz"))
    m = DISTS(load_weights=False)
    for tag in ("pub", "pert"):
        m.alpha.data = torch.from_numpy(g[f"proj_{tag}_in_alpha"]).view(1, -1, 1, 1).clone()
        m.beta.data = torch.from_numpy(g[f"proj_{tag}_in_beta"]).view(1, -1, 1, 1).clone()
        m.project_weights()
        assert np.abs(m.alpha.data.numpy().reshape(-1) - g[f"proj_{tag}_alpha"]).max() <= 1e-9
        assert np.abs(m.beta.data.numpy().reshape(-1) - g[f"proj_{tag}_beta"]).max() <= 1e-9


@pytest.mark.parametrize("mode", ["linear", "sqrt", "softmax", "softmax+sqrt"])
def test_mode_model_fit_matches_reference(mode):
    """nerf_qa/model.py:22-47: the least-squares initialisation of the head per wandb.config.mode."""
    import pandas as pd
    from nerf_qa_amd import config as cfgmod
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_original import DISTS as DO
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt_softmax import DISTS as DS
    from nerf_qa_amd.model import NeRFQAModel
    g = np.load(os.path.join(GOLDEN, "variants_64x64.npz"))
    c = cfgmod.config()
    c.mode, c.weight_lower_bound, c.alpha_beta_ratio, c.dists_weight_norm, c.detach_beta = mode, 1e-4, 1.0, "relu", "False"
    try:
        m = NeRFQAModel(pd.DataFrame({"DISTS": g["train_dists"], "MOS": g["train_mos"]}))
        got = np.array([m.dists_weight.item(), m.dists_bias.item()])
        want = g["mode_" + mode.replace("+", "_") + "_params"]
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6), (got, want)
        assert isinstance(m.dists_model, DS if "softmax" in mode else DO)
    finally:
        c.mode, c.weight_lower_bound, c.dists_weight_norm = "linear", 0.0, "off"


def test_video_columns_match_reference_expressions():
    """prep.py:191-198 / test2_prep.py:123-125,158-168 on synthetic float32 score vectors: float32 statistics,
    frame bias and its '{:.6e}' serialisation, batch count."""
    from nerf_qa_amd import video
    g = np.load(os.path.join(GOLDEN, "video_columns.npz"))
    for k in range(4):
        s = g[f"v{k}_scores"]
        c = video.video_columns("DISTS", s)
        got = np.array([c["DISTS"], c["DISTS_std"], c["DISTS_min"], c["DISTS_max"]])
        assert got.dtype == np.float32 and np.array_equal(got, g[f"v{k}_cols"])
        assert list(video.video_columns("A-DISTS", s, "_square")) == ["A-DISTS_square", "A-DISTS_square_std",
                                                                      "A-DISTS_square_min", "A-DISTS_square_max"]
        b = video.frame_bias(s)
        assert b.dtype == np.float32 and np.array_equal(b, g[f"v{k}_bias"])
        assert video.to_str(b) == str(g[f"v{k}_bias_str"])
        assert -(-len(s) // 8) == int(g[f"v{k}_batches"])
    assert video.to_str(np.array([0.1, 0.2], np.float32)) == "['1.000000e-01', '2.000000e-01']"


def test_csv_writer_layout(tmp_path):
    """The score table gets the reference's column names in the reference's order and goes out through
    DataFrame.to_csv with the index column (test2_prep.py:183-193,512); float32 cells print as pandas prints
    them in the reference's own scores_aspect.csv (shortest float32 repr)."""
    import pandas as pd
    from nerf_qa_amd import video
    df = pd.DataFrame({"distorted_folder": ["a", "b"], "reference_folder": ["r", "r"]})
    s = [np.array([0.25, 0.5, 0.75], np.float32), np.array([0.125], np.float32)]
    rows = []
    for v in s:
        r = {}
        r.update(video.video_columns("A-DISTS", v))
        r.update(video.video_columns("DISTS", v * 0.5))
        r.update({"frame_count": 1, "frame_bias_adists": video.to_str(video.frame_bias(v)),
                  "frame_bias_dists": video.to_str(video.frame_bias(v * 0.5))})
        rows.append(r)
    video.add_video_columns(df, rows)
    assert list(df.columns) == ["distorted_folder", "reference_folder", *video.COLUMN_ORDER]
    p = tmp_path / "scores.csv"
    video.write_scores_csv(df, str(p))
    lines = open(p).read().splitlines()
    assert lines[0] == ",distorted_folder,reference_folder," + ",".join(video.COLUMN_ORDER)
    assert lines[1].startswith("0,a,r,0.5,0.25,0.20412415,0.10206208,0.25,0.125,0.75,0.375,1,\"['2.500000e-01', ")
    sq = video.add_video_columns(pd.DataFrame({"x": [0]}), [video.video_columns("DISTS", s[0], "_square")], "_square")
    assert list(sq.columns) == ["x", "DISTS_square", "DISTS_square_std", "DISTS_square_min", "DISTS_square_max"]


def test_adists_as_loss_never_returns_a_silent_graphless_scalar():
    """as_loss=True (the reference's default) runs WITH autograd in the reference (ADISTS.py:139-141).  Since round 4 the
    drop-in has that gradient (HIP pyramid backward + the head in torch operations, tests/test_gpu_backward.py); on CPU
    tensors it refuses like every other entry point -- there is no CPU fallback -- instead of returning a scalar
    without a graph."""
    import nerf_qa_amd
    from nerf_qa_amd.ADISTS import ADISTS
    m = ADISTS()
    x = torch.rand(1, 3, 32, 32, requires_grad=True)
    with pytest.raises(nerf_qa_amd.NqaError):
        m(x, torch.rand(1, 3, 32, 32))
    with pytest.raises(nerf_qa_amd.NqaError):
        m(torch.rand(1, 3, 32, 32), x, as_loss=True)


def test_workspace_is_per_stream_and_capped():
    """ops.Workspace: one grow-only buffer per (device, stream) -- on the CPU one key -- and never more than MAX_STREAMS of
    them (least recently used first out)."""
    from nerf_qa_amd import ops
    w, cpu = ops.Workspace(), torch.device("cpu")
    a = w.get(100, cpu)
    assert w.get(50, cpu) is a and a.numel() == 256 and len(w.bufs) == 1
    b = w.get(4096, cpu)
    assert b.numel() == 4096 and len(w.bufs) == 1 and w.get(4096, cpu) is b
    for k in range(20):  # (stand-ins for twenty short-lived streams)
        w.bufs[("cpu", 1000 + k)] = torch.empty(1, dtype=torch.uint8)
        w.bufs.pop(("cpu", 0), None)
        w.get(256, cpu)
    assert len(w.bufs) <= ops.Workspace.MAX_STREAMS + 1
