"""How the fast modes behave on nearly flat frames (one colour +- a small smooth variation) against a render with floaters --
the content family that carries the stress run's worst case: |mode - f32s| of the DISTS score by amplitude of the variation,
with the frames' pixel variance (development aid, GPU box).  usage: python tools/gpu_flat_frames.py"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
H, W, N = 1080, 1440, 24
for gain, mode in ((1.0, "f16"), (1.3, "f32m")):
    spec = "synth:1234" if gain == 1.0 else f"synth:1234:{gain}"
    fast, exact = DISTS(precision=mode, vgg16_path=spec).to(dev).eval(), DISTS(precision="f32s", vgg16_path=spec).to(dev).eval()
    auto = DISTS(vgg16_path=spec).to(dev).eval()  # the shipped default: the same rung at this size, behind the flat-frame guard
    for amp in (0.2, 0.05, 0.02, 0.01, 0.005, 0.002, 0.0):
        errs, vars_, aerr = [], [], []
        for i in range(N):
            low = F.interpolate(torch.rand(1, 3, H // 16, W // 16, device=dev, generator=g), size=(H, W), mode="bilinear", align_corners=False)
            field = F.interpolate(torch.rand(1, 1, 5, 6, device=dev, generator=g), size=(H, W), mode="bicubic", align_corners=False)
            col = torch.rand(1, 3, 1, 1, device=dev, generator=g) * 0.6 + 0.2
            xs = (col + amp * (low - 0.5)).clamp_(0, 1)
            blob = ((field - 0.85) * 60.0).clamp(0, 1)
            ys = ((1 - blob) * xs + blob * torch.rand(1, 3, 1, 1, device=dev, generator=g)).clamp_(0, 1)
            with torch.no_grad():
                ex = float(exact(xs, ys))
                errs.append(abs(float(fast(xs, ys)) - ex))
                aerr.append(abs(float(auto(xs, ys)) - ex))
            vars_.append(float(xs.var(dim=(2, 3)).mean()))
        print(f"gain {gain} {mode}: variation +-{amp / 2:.4f}: pixel variance of x {np.median(vars_):.1e}; |{mode} - f32s| max {max(errs):.2e} median {np.median(errs):.2e}; |auto - f32s| max {max(aerr):.2e}", flush=True)
