"""Where does the 16-bit score error come from?  (development aid, GPU box only)

Reference = the f32 HIP path (validated against the CPU oracle to 1e-7).  Compared:
  f16          : f16 activations + f16 weights
  f32/w16      : f32 kernels fed weights that were rounded to f16 first  (weight rounding only)
  bf16, f32/wb16 likewise
"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
convs = synth.vgg16_weights(1234)
ab = np.load("nerf_qa_amd/data/dists_alpha_beta.npz")
alpha, beta = torch.from_numpy(ab["alpha"]).to(dev), torch.from_numpy(ab["beta"]).to(dev)


def rounded(convs, dt):
    return [(torch.from_numpy(w).to(dt).float().numpy(), b) for w, b in convs]


packs = {
    "f32": (ops.pack_vgg_weights(convs, "f32").to(dev), "f32"),
    "f16": (ops.pack_vgg_weights(convs, "f16").to(dev), "f16"),
    "bf16": (ops.pack_vgg_weights(convs, "bf16").to(dev), "bf16"),
    "f32/w16": (ops.pack_vgg_weights(rounded(convs, torch.float16), "f32").to(dev), "f32"),
    "f32/wb16": (ops.pack_vgg_weights(rounded(convs, torch.bfloat16), "f32").to(dev), "f32"),
}


def score(x, y, key):
    p, prec = packs[key]
    return ops.dists_score(*ops.dists_forward(x, y, p, prec), alpha, beta)


for (h, w) in [(32, 32), (64, 96), (128, 128), (256, 256), (512, 512), (1080, 1920)]:
    nb = 8 if h * w <= 512 * 512 else 2
    seeds = list(range(200, 200 + nb))
    xn, yn = synth.frame_batch(seeds, h, w)
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    ref = score(x, y, "f32")
    line = f"{h}x{w} n={nb} score range [{ref.min().item():.4f},{ref.max().item():.4f}]"
    for key in ("f16", "f32/w16", "bf16", "f32/wb16"):
        d = (score(x, y, key) - ref).abs()
        line += f" | {key}: max {d.max().item():.2e} mean {d.mean().item():.2e}"
    print(line, flush=True)
