#!/usr/bin/env python3
"""Pin the oracle against the reference itself and freeze golden vectors.

AUTHORING-CONTAINER ONLY (needs /root/reference).  Run:  python -m oracle.make_goldens [part ...]
parts: small (seconds), full (BASELINE.json's full-size configs, ~15 min of CPU), variants, video, nerf (NeRF-like
content families, ~1 min); default all.  (full1080g13: only round 4's addition to `full`.)

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


def import_reference(gain=1.0):
    """The reference's own classes, with the stand-in torchvision serving synth.vgg16_weights(WEIGHT_SEED, gain)."""
    if os.path.join(ROOT, "oracle", "_standin") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_standin"))
        sys.path.insert(1, REF)
    import torchvision.models as tvm
    np_convs = synth.vgg16_weights(WEIGHT_SEED, gain)
    tvm.WEIGHT_PROVIDER = lambda: np_convs
    from nerf_qa.ADISTS import ADISTS as RefADISTS
    from nerf_qa.DISTS_pytorch.DISTS_pt import DISTS as RefDISTS
    return RefDISTS, RefADISTS, np_convs


def published_alpha_beta():
    ab = torch.load(os.path.join(REF, "nerf_qa", "DISTS_pytorch", "weights.pt"))
    return ab["alpha"].float(), ab["beta"].float()


# ---- BASELINE.json's full-size configs, from the reference itself ------------------------------------
# configs[1]: B=32 of 256x256; configs[2]: B=8 of 1080p; configs[4]: A-DISTS at 1080p.  The reference is run one
# pair at a time (its result for a pair does not depend on the batch it sits in; 1080p pairs need several GB each);
# the GPU tests place these pairs at chosen slots of a full batch.  Two weight sets: the unit-gain stand-in every
# other golden uses, and gain 1.6, whose activations grow with depth as ImageNet weights' do (relu5_3: mean ~150,
# max ~1700 at 1080p), so 16-bit range, the L2-pool squaring and the statistics are stressed;
# gain 1.3 (relu5_3 mean ~10, max ~120 -- the ImageNet-like magnitude) is added for the cheap 256x256 batch.
FULL_DISTS = [
    ("b32_256", 256, 256, tuple(range(200, 232)), None, (1.0, 1.3, 1.6)),
    ("1080p", 1080, 1920, (300, 301, 302), ("blur", "noise10", "indep"), (1.0, 1.3, 1.6)),  # (1.3 added in round 4)
]
FULL_ADISTS = [
    ("b8_256", 256, 256, tuple(range(240, 248)), None, (1.0, 1.6)),
    ("1080p", 1080, 1920, (310, 311), ("blur", "noise10"), (1.0, 1.6)),
]


def gain_tag(gain):
    return "" if gain == 1.0 else "_g%d" % round(gain * 10)


# ---- NeRF-render-like content (round 4): constant backgrounds, smooth frames, floaters ---------
# 12 pairs of 256x256 per weight set (three of each synth.NERF_KINDS family), DISTS and A-DISTS, from the reference.
NERF_SEEDS = tuple(range(400, 412))


def nerf_goldens(gold, only_gain=None):
    alpha, beta = published_alpha_beta()
    for gain in (1.0, 1.3, 1.6):
        if only_gain is not None and gain != only_gain:
            continue
        RefDISTS, RefADISTS, np_convs = import_reference(gain)
        convs = dists_oracle.convs_from_numpy(np_convs)
        ref_d = RefDISTS(load_weights=False).eval()
        ref_d.alpha.data, ref_d.beta.data = alpha.clone(), beta.clone()
        ref_a = RefADISTS().eval()
        kinds = [synth.NERF_KINDS[i % 4] for i in range(len(NERF_SEEDS))]
        xn, yn = synth.frame_batch(NERF_SEEDS, 256, 256, kinds)
        x, y = torch.from_numpy(xn), torch.from_numpy(yn)
        with torch.no_grad():
            r = ref_d(x, y)
            f0, f1 = dists_oracle.vgg_pyramid(x, convs), dists_oracle.vgg_pyramid(y, convs)
            s1, s2 = dists_oracle.dists_stats(f0, f1)
            o = dists_oracle.dists_score(s1, s2, alpha, beta)
            ra = ref_a(x, y, as_loss=False)
            oa = adists_oracle.adists(x, y, convs, as_loss=False)
        d, da = (o - r).abs().max().item(), (oa - ra).abs().max().item()
        assert d <= 2e-6 and da <= 2e-6, f"nerf families gain {gain}: oracle differs from the reference by {d} / {da}"
        # dead channels per tap on these frames (what the family is for): fraction of (pair, channel) with a zero variance
        dead = [float(((f.flatten(2).var(2, unbiased=False) < 1e-12).float().mean())) for f in f0[1:]]
        np.savez(os.path.join(gold, f"nerf_256{gain_tag(gain)}.npz"), h=256, w=256, seeds=np.array(NERF_SEEDS),
                 kinds=np.array(kinds), weight_seed=WEIGHT_SEED, weight_gain=gain, score=r.numpy(), s1=s1.numpy(),
                 s2=s2.numpy(), adists=ra.numpy(), dead_frac=np.array(dead))
        print(f"nerf families gain {gain}: DISTS {np.round(r.numpy(), 4).tolist()} |oracle-ref|={d:.1e}; "
              f"A-DISTS {np.round(ra.numpy(), 4).tolist()} |oracle-ref|={da:.1e}; dead-channel fraction per tap {np.round(dead, 3).tolist()}",
              flush=True)


def fullsize_goldens(gold, only=None):
    alpha, beta = published_alpha_beta()
    import time
    for gain in (1.0, 1.3, 1.6):
        if only and gain != only[1]:
            continue
        RefDISTS, RefADISTS, np_convs = import_reference(gain)
        convs = dists_oracle.convs_from_numpy(np_convs)
        ref_d = RefDISTS(load_weights=False).eval()
        ref_d.alpha.data, ref_d.beta.data = alpha.clone(), beta.clone()
        ref_a = RefADISTS().eval()
        for name, h, w, seeds, kinds, gains in FULL_DISTS:
            if gain not in gains or (only and name != only[0]):
                continue
            scores, s1s, s2s, summ = [], [], [], []
            for i, seed in enumerate(seeds):
                kind = kinds[i % len(kinds)] if kinds else synth.KINDS[i % 4]
                xn, yn = synth.frame_pair(seed, h, w, kind)
                x, y = torch.from_numpy(xn), torch.from_numpy(yn)
                t0 = time.time()
                with torch.no_grad():
                    r = ref_d(x, y)
                    f0, f1 = dists_oracle.vgg_pyramid(x, convs), dists_oracle.vgg_pyramid(y, convs)
                    s1, s2 = dists_oracle.dists_stats(f0, f1)
                    o = dists_oracle.dists_score(s1, s2, alpha, beta)
                d = (o - r).abs().max().item()
                assert d <= 2e-6, f"{name} seed {seed}: oracle differs from the reference by {d}"
                scores.append(r.numpy()), s1s.append(s1.numpy()), s2s.append(s2.numpy())
                summ.append([[f.mean().item(), f.abs().max().item()] for f in f0])
                print(f"DISTS {name}{gain_tag(gain)} seed {seed} {kind}: ref={r.item():.6f} |oracle-ref|={d:.1e} "
                      f"relu5_3 mean/max={summ[-1][5][0]:.3g}/{summ[-1][5][1]:.3g} ({time.time() - t0:.0f}s)", flush=True)
                del f0, f1
            kk = [kinds[i % len(kinds)] if kinds else synth.KINDS[i % 4] for i in range(len(seeds))]
            np.savez(os.path.join(gold, f"full_dists_{name}{gain_tag(gain)}.npz"), h=h, w=w, seeds=np.array(seeds),
                     kinds=np.array(kk), weight_seed=WEIGHT_SEED, weight_gain=gain, score=np.concatenate(scores),
                     s1=np.concatenate(s1s), s2=np.concatenate(s2s), feat_x=np.array(summ))
        for name, h, w, seeds, kinds, gains in FULL_ADISTS:
            if gain not in gains or only:
                continue
            scores = []
            for i, seed in enumerate(seeds):
                kind = kinds[i % len(kinds)] if kinds else synth.KINDS[i % 4]
                xn, yn = synth.frame_pair(seed, h, w, kind)
                x, y = torch.from_numpy(xn), torch.from_numpy(yn)
                t0 = time.time()
                with torch.no_grad():
                    r = ref_a(x, y, as_loss=False)
                    o = adists_oracle.adists(x, y, convs, as_loss=False)
                d = (o - r).abs().max().item()
                assert d <= 2e-6, f"{name} seed {seed}: A-DISTS oracle differs from the reference by {d}"
                scores.append(r.numpy())
                print(f"ADISTS {name}{gain_tag(gain)} seed {seed} {kind}: ref={r.item():.6f} |oracle-ref|={d:.1e} "
                      f"({time.time() - t0:.0f}s)", flush=True)
            kk = [kinds[i % len(kinds)] if kinds else synth.KINDS[i % 4] for i in range(len(seeds))]
            np.savez(os.path.join(gold, f"full_adists_{name}{gain_tag(gain)}.npz"), h=h, w=w, seeds=np.array(seeds),
                     kinds=np.array(kk), weight_seed=WEIGHT_SEED, weight_gain=gain, score=np.concatenate(scores))


# ---- the video harness's column arithmetic (prep.py:191-198, test2_prep.py:123-125,158-168) --------------
def video_goldens(gold):
    """The reference's scripts cannot be imported (they run on import against private datasets), so the few numpy
    expressions they apply to the per-frame score arrays are evaluated here, verbatim, on synthetic float32 score
    vectors; nerf_qa_amd.video must reproduce the results to the bit / to the character."""
    def to_str(array):                                   # test2_prep.py:123-125
        array = ['{:.6e}'.format(num) for num in array]
        return str(array)
    rng = np.random.default_rng(7)
    out = {}
    for k, n in enumerate((1, 5, 200, 1001)):
        batches = [rng.uniform(0.0, 0.4, m).astype(np.float32) for m in ([8] * (n // 8) + ([n % 8] if n % 8 else []))]
        frame = np.concatenate(batches)                  # test2_prep.py:156-157
        out[f"v{k}_scores"] = frame
        out[f"v{k}_cols"] = np.array([np.mean(frame), np.std(frame), np.min(frame), np.max(frame)])  # :158-165
        assert out[f"v{k}_cols"].dtype == np.float32
        bias = np.mean(frame) - frame                    # :167-168
        out[f"v{k}_bias"] = bias
        out[f"v{k}_bias_str"] = np.array(to_str(bias))   # :179-180
        out[f"v{k}_batches"] = np.array(len(batches))    # :181 len(frames_data)
    np.savez(os.path.join(gold, "video_columns.npz"), **out)
    print("video column goldens:", {k: out[f"v{k}_cols"] for k in range(4)})


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
    # nerf_qa/model.py:22-56: the `wandb.config.mode` variant of NeRFQAModel
    from nerf_qa.model import NeRFQAModel as RefModeModel
    cfg.weight_lower_bound, cfg.alpha_beta_ratio, cfg.dists_weight_norm, cfg.detach_beta = 1e-4, 1.0, "relu", "False"
    for mode in ("linear", "sqrt", "softmax", "softmax+sqrt"):
        cfg.mode = mode
        M = RefModeModel(df).eval()
        with torch.no_grad():
            scores, ds = M(x, y)
        key = mode.replace("+", "_")
        out[f"mode_{key}_params"] = np.array([M.dists_weight.item(), M.dists_bias.item()])
        out[f"mode_{key}_scores"], out[f"mode_{key}_dists"] = scores.numpy(), ds.numpy()
        print(f"model.py mode {mode}: params {out[f'mode_{key}_params']} scores {scores.numpy()}")
    torch.load = real_load
    # canonical DISTS.project_weights (DISTS_pt.py:82-89) on the published and on perturbed alpha/beta
    from nerf_qa.DISTS_pytorch.DISTS_pt import DISTS as RefDISTS
    alpha, beta = published_alpha_beta()
    g = torch.Generator().manual_seed(3)
    for tag, (a0, b0) in {"pub": (alpha, beta), "pert": (alpha + 0.01 * torch.randn(alpha.shape, generator=g),
                                                         beta - 0.002 * torch.rand(beta.shape, generator=g))}.items():
        m = RefDISTS(load_weights=False)
        m.alpha.data, m.beta.data = a0.clone(), b0.clone()
        m.project_weights()
        out[f"proj_{tag}_in_alpha"], out[f"proj_{tag}_in_beta"] = a0.numpy().reshape(-1), b0.numpy().reshape(-1)
        out[f"proj_{tag}_alpha"], out[f"proj_{tag}_beta"] = m.alpha.data.numpy().reshape(-1), m.beta.data.numpy().reshape(-1)
    np.savez(os.path.join(gold, "variants_64x64.npz"), **out)
    prepare_image_goldens(gold)


def prepare_image_goldens(gold):
    """prepare_image of DISTS_pt (:210-217) and DISTS_pt_original (:140-144) run on seeded PIL images (the
    stand-in transforms are Pillow-backed): output shapes and pixels for every resize policy."""
    from PIL import Image
    from nerf_qa.DISTS_pytorch.DISTS_pt import prepare_image as ref_prep
    from nerf_qa.DISTS_pytorch.DISTS_pt_original import prepare_image as ref_prep_orig
    out = {}
    for k, (h, w) in enumerate(((300, 420), (540, 300), (200, 640), (256, 256), (1080, 1920))):
        arr = (synth.uniform(900 + k, h * w * 3).reshape(h, w, 3) * 256).astype(np.uint8)
        img = Image.fromarray(arr, "RGB")
        calls = {"sq": lambda: ref_prep(img), "keep": lambda: ref_prep(img, resize=True, keep_aspect_ratio=True),
                 "none": lambda: ref_prep(img, resize=False), "orig": lambda: ref_prep_orig(img),
                 "orig_none": lambda: ref_prep_orig(img, resize=False)}
        for tag, fn in calls.items():
            t = fn()
            out[f"p{k}_{tag}_shape"] = np.array(t.shape)
            out[f"p{k}_{tag}_sum"] = np.array(t.double().sum().item())
            out[f"p{k}_{tag}_u8"] = (t[0, :, ::37, ::41] * 255).round().to(torch.uint8).numpy()  # sparse sample
        out[f"p{k}_hw"] = np.array([h, w])
    np.savez(os.path.join(gold, "prepare_image.npz"), **out)
    print("prepare_image goldens:", {k: out[k].tolist() for k in out if k.endswith("orig_shape")})


def feat_summary(feats):
    return np.array([[f.mean().item(), f.abs().mean().item(), f.abs().max().item()] for f in feats], np.float64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    parts = set(sys.argv[1:]) or {"small", "full", "variants", "video", "nerf"}
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    if "video" in parts:
        video_goldens(gold)
    if "full" in parts:
        fullsize_goldens(gold)
    if "full1080g13" in parts:  # (only the file round 4 added: three 1080p pairs at gain 1.3, ~6 min of CPU)
        fullsize_goldens(gold, only=("1080p", 1.3))
    if "nerf" in parts:
        nerf_goldens(gold)
    if "variants" in parts:
        import_reference()
        variant_goldens(gold)
    if "small" in parts:
        small_goldens(gold)
    print("goldens written to", gold)


def small_goldens(gold):
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

    # weight fingerprint so a drift of the generator is caught on the GPU box too
    fp = np.array([[float(np.abs(w_).sum()), float(b_.sum())] for w_, b_ in np_convs])
    np.savez(os.path.join(gold, "vgg_fingerprint.npz"), weight_seed=WEIGHT_SEED, fp=fp)


if __name__ == "__main__":
    main()
