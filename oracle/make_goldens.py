#!/usr/bin/env python3
"""Pin the oracle against the reference itself and freeze golden vectors.

AUTHORING-CONTAINER ONLY (needs /root/reference).  Run:  python -m oracle.make_goldens

What it does
  1. puts oracle/_standin (a local `torchvision` stand-in; the real package is not
     installed) ahead of /root/reference on sys.path and imports the reference's own
     nerf_qa.DISTS_pytorch.DISTS_pt.DISTS and nerf_qa.ADISTS.ADISTS;
  2. injects the deterministic VGG weights of nerf_qa_amd.synth into the stand-in
     and assigns alpha/beta from the reference's weights.pt (what DISTS_pt.py:63,79-80
     would do with sys.prefix/weights.pt);
  3. runs the reference on seeded synthetic frame pairs, runs oracle/ on the same
     inputs, REQUIRES them to agree (max-abs 2e-6 on scores; features bit-exact),
  4. writes tests/golden/*.npz (inputs are regenerated from seeds; only outputs and
     small summaries are stored) and nerf_qa_amd/data/dists_alpha_beta.npz (the
     published DISTS alpha/beta as a data fixture).

The reference source never enters the repo; only its numeric outputs do.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from nerf_qa_amd import synth  # noqa: E402
from oracle import adists_oracle, dists_oracle  # noqa: E402

# (name, H, W, seeds, kinds)
DISTS_CASES = [
    ("64x64", 64, 64, (0, 1, 2, 3), None),
    ("97x131", 97, 131, (10, 11, 12, 13), None),
    ("256x256", 256, 256, (20, 21, 22, 23), None),
    ("256x341", 256, 341, (30, 31), ("blur", "noise10")),
    ("20x20", 20, 20, (40, 41), ("noise10", "indep")),
    ("same64", 64, 64, (50,), ("same",)),
]
ADISTS_CASES = [
    ("64x64", 64, 64, (0, 1), ("noise10", "blur")),
    ("97x131", 97, 131, (10, 11), ("noise02", "indep")),
    ("256x256", 256, 256, (20, 21), ("blur", "noise10")),
    ("20x20", 20, 20, (40, 41), ("noise10", "indep")),       # every stage takes the global fallback
    ("352x336", 352, 336, (60,), ("blur",)),                  # stage 5 is 22x21: windowed everywhere
]
AMAP_CASES = ("64x64", "97x131", "20x20")  # as_map=True goldens (the map of column j = 0; all columns are equal)
WEIGHT_SEED = 1234


def import_reference():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_standin"))
    sys.path.insert(1, REF)
    import torchvision.models as tvm
    np_convs = synth.vgg16_weights(WEIGHT_SEED)
    tvm.WEIGHT_PROVIDER = lambda: np_convs
    from nerf_qa.ADISTS import ADISTS as RefADISTS
    from nerf_qa.DISTS_pytorch.DISTS_pt import DISTS as RefDISTS
    return RefDISTS, RefADISTS, np_convs


VARIANT_CONFIGS = [  # (weight_lower_bound, alpha_beta_ratio, dists_weight_norm, detach_beta)
    (0.0, 1.0, "off", "False"),
    (2e-4, 2.0, "relu", "False"),
    (1e-4, 0.5, "relu+w_sum_detach", "True"),
]


def variant_goldens(gold):
    """The training-time variants (DISTS_pt_original / _softmax) and the NeRFQAModel head, from the reference
    itself: scores under several run configs, project_weights, the fitted head parameters and its outputs."""
    import pandas as pd
    import wandb  # oracle/_standin/wandb: the attribute bag the reference reads its hyper-parameters from
    from nerf_qa.DISTS_pytorch.DISTS_pt_original import DISTS as RefOrig
    from nerf_qa.DISTS_pytorch.DISTS_pt_softmax import DISTS as RefSoft
    from nerf_qa.model_stats import NeRFQAModel as RefModel
    # the reference loads sys.prefix/weights.pt (where its packaging puts the file): serve the in-repo copy
    real_load = torch.load

    def load(path, *a, **k):
        if os.path.basename(str(path)) == "weights.pt":
            path = os.path.join(REF, "nerf_qa", "DISTS_pytorch", "weights.pt")
        return real_load(path, *a, **k)
    torch.load = load
    cfg = wandb.config
    cfg.subjective_score_type, cfg.regression_type = "MOS", "linear"
    xn, yn = synth.frame_batch((11, 12), 64, 64)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    out = {"h": 64, "w": 64, "seeds": np.array((11, 12)), "weight_seed": WEIGHT_SEED,
           "configs": np.array([[c[0], c[1]] for c in VARIANT_CONFIGS]),
           "norms": np.array([c[2] for c in VARIANT_CONFIGS]), "detach": np.array([c[3] for c in VARIANT_CONFIGS])}
    for i, (lb, ratio, norm, det) in enumerate(VARIANT_CONFIGS):
        cfg.weight_lower_bound, cfg.alpha_beta_ratio, cfg.dists_weight_norm, cfg.detach_beta = lb, ratio, norm, det
        m = RefOrig().eval()
        with torch.no_grad():
            s = m(x, y)
            one = m(x[:1], y[:1])
        assert s.shape == (2,) and one.dim() == 0
        m.project_weights()
        with torch.no_grad():
            sp = m(x, y)
        out[f"orig{i}_score"], out[f"orig{i}_one"], out[f"orig{i}_projected"] = s.numpy(), one.numpy(), sp.numpy()
        out[f"orig{i}_alpha"], out[f"orig{i}_beta"] = m.alpha.data.numpy().reshape(-1), m.beta.data.numpy().reshape(-1)
        print(f"variant original cfg{i}: {s.numpy()} -> projected {sp.numpy()}")
    cfg.dists_weight_norm, cfg.detach_beta = "softmax", "False"
    ms = RefSoft().eval()
    with torch.no_grad():
        out["soft_score"] = ms(x, y).numpy()
    rng = np.random.default_rng(0)
    d = rng.uniform(0.05, 0.4, 40)
    mos = 5 - 8 * d + 0.01 * rng.standard_normal(40)
    out["train_dists"], out["train_mos"] = d, mos
    df = pd.DataFrame({"DISTS": d, "MOS": mos})
    cfg.weight_lower_bound, cfg.alpha_beta_ratio, cfg.dists_weight_norm, cfg.detach_beta = 1e-4, 1.0, "relu", "False"
    for kind in ("linear", "sqrt", "logistic"):
        cfg.regression_type = kind
        M = RefModel(df).eval()
        with torch.no_grad():
            scores, ds = M(x, y)
            ent = M.entropy_loss()
        params = [M.b1, M.b2, M.b3, M.b4] if kind == "logistic" else [M.dists_weight, M.dists_bias]
        out[f"head_{kind}_params"] = np.array([p.item() for p in params])
        out[f"head_{kind}_scores"], out[f"head_{kind}_dists"], out[f"head_{kind}_entropy"] = \
            scores.numpy(), ds.numpy(), np.array(ent.item())
        print(f"head {kind}: params {out[f'head_{kind}_params']} scores {scores.numpy()}")
    torch.load = real_load
    np.savez(os.path.join(gold, "variants_64x64.npz"), **out)


def feat_summary(feats):
    return np.array([[f.mean().item(), f.abs().mean().item(), f.abs().max().item()] for f in feats], np.float64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    RefDISTS, RefADISTS, np_convs = import_reference()
    convs = dists_oracle.convs_from_numpy(np_convs)
    ab = torch.load(os.path.join(REF, "nerf_qa", "DISTS_pytorch", "weights.pt"))
    alpha, beta = ab["alpha"].float(), ab["beta"].float()
    os.makedirs(os.path.join(ROOT, "nerf_qa_amd", "data"), exist_ok=True)
    np.savez(os.path.join(ROOT, "nerf_qa_amd", "data", "dists_alpha_beta.npz"),
             alpha=alpha.numpy().reshape(-1), beta=beta.numpy().reshape(-1))

    ref_d = RefDISTS(load_weights=False).eval()
    ref_d.alpha.data = alpha.clone()
    ref_d.beta.data = beta.clone()
    ref_a = RefADISTS().eval()
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)

    for name, h, w, seeds, kinds in DISTS_CASES:
        xn, yn = synth.frame_batch(seeds, h, w, kinds)
        x, y = torch.from_numpy(xn), torch.from_numpy(yn)
        with torch.no_grad():
            r_score = ref_d(x, y)
            r_f0, r_f1 = ref_d.forward_once(x), ref_d.forward_once(y)
            r_ff = ref_d.forward_from_feats(r_f0, r_f1)
            r_avg = ref_d(x, y, batch_average=True)
        o_f0, o_f1 = dists_oracle.vgg_pyramid(x, convs), dists_oracle.vgg_pyramid(y, convs)
        for a, b in zip(r_f0 + r_f1, o_f0 + o_f1):
            assert torch.equal(a, b), f"{name}: oracle pyramid differs from reference"
        s1, s2 = dists_oracle.dists_stats(o_f0, o_f1)
        o_score = dists_oracle.dists_score(s1, s2, alpha, beta)
        d = (o_score - r_score).abs().max().item()
        assert d <= 2e-6, f"{name}: oracle score differs from reference by {d}"
        assert (r_ff - r_score).abs().max().item() <= 1e-6
        o_avg = dists_oracle.dists(x, y, convs, alpha, beta, batch_average=True)
        assert abs(o_avg.item() - r_avg.item()) <= 2e-6
        print(f"DISTS {name:8s} ref={r_score.numpy()} |oracle-ref|={d:.2e}")
        np.savez(os.path.join(gold, f"dists_{name}.npz"),
                 h=h, w=w, seeds=np.array(seeds), kinds=np.array(kinds if kinds else synth.KINDS[:4]),
                 weight_seed=WEIGHT_SEED, score=r_score.numpy(), score_avg=r_avg.numpy(),
                 s1=s1.numpy(), s2=s2.numpy(), feat_x=feat_summary(r_f0), feat_y=feat_summary(r_f1))

    for name, h, w, seeds, kinds in ADISTS_CASES:
        xn, yn = synth.frame_batch(seeds, h, w, kinds)
        x, y = torch.from_numpy(xn), torch.from_numpy(yn)
        with torch.no_grad():
            r_score = ref_a(x, y, as_loss=False)
            r_loss = ref_a(x, y, as_loss=True)
        o_score = adists_oracle.adists(x, y, convs, as_loss=False)
        o_loss = adists_oracle.adists(x, y, convs, as_loss=True)
        d = (o_score - r_score).abs().max().item()
        assert d <= 2e-6, f"{name}: A-DISTS oracle differs from reference by {d}"
        assert abs(o_loss.item() - r_loss.item()) <= 2e-6
        print(f"ADISTS {name:8s} ref={r_score.numpy()} |oracle-ref|={d:.2e}")
        np.savez(os.path.join(gold, f"adists_{name}.npz"),
                 h=h, w=w, seeds=np.array(seeds), kinds=np.array(kinds), weight_seed=WEIGHT_SEED,
                 score=r_score.numpy(), loss=r_loss.numpy())
        if name in AMAP_CASES:  # as_map=True (ADISTS.py:188-193): (B,B,H,W), out[i,j] = map[i]
            with torch.no_grad():
                r_map = ref_a(x, y, as_loss=False, as_map=True)
            o_map = adists_oracle.adists(x, y, convs, as_map=True)
            assert r_map.shape == o_map.shape == (len(seeds), len(seeds), h, w)
            dm = (o_map - r_map).abs().max().item()
            assert dm <= 2e-6, f"{name}: A-DISTS map oracle differs from reference by {dm}"
            assert all(torch.equal(r_map[:, j], r_map[:, 0]) for j in range(r_map.shape[1]))
            print(f"ADISTS map {name:8s} shape={tuple(r_map.shape)} |oracle-ref|={dm:.2e}")
            np.savez(os.path.join(gold, f"amap_{name}.npz"),
                     h=h, w=w, seeds=np.array(seeds), kinds=np.array(kinds), weight_seed=WEIGHT_SEED,
                     shape=np.array(r_map.shape), map=r_map[:, 0].numpy())

    variant_goldens(gold)

    # weight fingerprint so a drift of the generator is caught on the GPU box too
    fp = np.array([[float(np.abs(w_).sum()), float(b_.sum())] for w_, b_ in np_convs])
    np.savez(os.path.join(gold, "vgg_fingerprint.npz"), weight_seed=WEIGHT_SEED, fp=fp)
    print("goldens written to", gold)


if __name__ == "__main__":
    main()
