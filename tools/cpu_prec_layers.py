"""Which layers' 16-bit WEIGHT rounding carries the f16 score error?  (development aid, CPU only)

Emulates the f16 HIP path on CPU PyTorch: activations rounded to f16 wherever the HIP path stores them (after every
conv+ReLU and every L2-pool; the normalised input), products and sums in float32.  A layer's weights are either
rounded to f16 (one MFMA per product) or kept as f16 hi + f16 lo (two MFMAs per product, ~22 bits = float32 here).
Compares against the committed golden scores of the imported reference (tests/golden/full_dists_b32_256*.npz).

usage: python tools/cpu_prec_layers.py [gain] [npairs] [mode ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_qa_amd import synth  # noqa: E402
from oracle import dists_oracle as do  # noqa: E402


def h(t):
    return t.half().float()


def pyramid(x, convs, wsplit, asplit=(), tap32=False):
    """wsplit: set of layer indices whose weights keep hi+lo; asplit: layers whose INPUT activation stays f32;
    tap32: the statistics read unrounded taps."""
    mean = torch.tensor(do.IMAGENET_MEAN).view(1, -1, 1, 1)
    std = torch.tensor(do.IMAGENET_STD).view(1, -1, 1, 1)
    a = (x - mean) / std
    feats = [x]
    li = 0
    for s, nconv in enumerate(do.STAGE_CONVS):
        if s > 0:
            a = do.l2pool(a)
        for _ in range(nconv):
            w, b = convs[li]
            ain = a if li in asplit else h(a)
            wl = w if li in wsplit else h(w)
            a = F.relu(F.conv2d(ain, wl, b, padding=1))
            li += 1
        feats.append(a if tap32 else h(a))  # taps are stored f16
    return feats


def main():
    gain = float(sys.argv[1]) if len(sys.argv) > 1 else 1.6
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    modes = sys.argv[3:] or ["none", "all"]
    tag = "" if gain == 1.0 else "_g%d" % round(gain * 10)
    g = np.load(os.path.join(ROOT, "tests/golden", f"full_dists_b32_256{tag}.npz"))
    convs = do.convs_from_numpy(synth.vgg16_weights(1234, gain))
    ab = np.load(os.path.join(ROOT, "nerf_qa_amd/data/dists_alpha_beta.npz"))
    alpha, beta = torch.from_numpy(ab["alpha"]), torch.from_numpy(ab["beta"])
    xs, ys = [], []
    for seed, kind in list(zip(g["seeds"], g["kinds"]))[:npairs]:
        x, y = synth.frame_pair(int(seed), int(g["h"]), int(g["w"]), str(kind))
        xs.append(torch.from_numpy(x))
        ys.append(torch.from_numpy(y))
    x, y = torch.cat(xs), torch.cat(ys)
    ref = g["score"][:npairs]
    for mode in modes:
        asp, tap32 = set(), False
        if "/" in mode:
            mode, extra = mode.split("/", 1)
            for e in extra.split("/"):
                if e == "t32":
                    tap32 = True
                elif e == "aall":
                    asp = set(range(13))
                elif e.startswith("a"):
                    asp = {int(t) for t in e[1:].split(",")}
        if mode.startswith("wonly"):
            ws, asp, tap32 = set(range(13)) - {int(mode[5:])}, set(range(13)), True
        elif mode.startswith("aonly"):
            ws, asp, tap32 = set(range(13)), set(range(13)) - {int(mode[5:])}, True
        elif mode == "none":
            ws = set()
        elif mode == "all":
            ws = set(range(13))
        elif mode.startswith("only"):
            ws = {int(t) for t in mode[4:].split(",")}
        elif mode.startswith("not"):
            ws = set(range(13)) - {int(t) for t in mode[3:].split(",")}
        else:
            raise SystemExit(mode)
        with torch.no_grad():
            f0, f1 = pyramid(x, convs, ws, asp, tap32), pyramid(y, convs, ws, asp, tap32)
            s1, s2 = do.dists_stats(f0, f1)
            sc = do.dists_score(s1, s2, alpha, beta).numpy()
        d = sc - ref
        print(f"gain {gain} w-split {sorted(ws)} a-f32 {sorted(asp)} tap32 {tap32}: max|d|={np.abs(d).max():.2e} rms={np.sqrt((d*d).mean()):.2e} "
              f"mean={d.mean():+.2e}", flush=True)


if __name__ == "__main__":
    main()
