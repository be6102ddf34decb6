"""How stable and how discriminating is DISTS' auto-precision calibration?  (development aid, GPU box)
For the three pinned stand-in weight sets: max / rms of |f16 - f32s| over calibration sets of several seeds and sizes."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch.DISTS_pt import _DATA, calibration_pairs  # noqa: E402

dev = torch.device("cuda:0")
ab = np.load(_DATA)
a, b = torch.from_numpy(ab["alpha"]).to(dev), torch.from_numpy(ab["beta"]).to(dev)
for gain in (1.0, 1.3, 1.6):
    m = DISTS(vgg16_path=f"synth:1234:{gain}", precision="f32s").to(dev).eval()
    p16, p32 = m._packed_weights(dev, "f16"), m._packed_weights(dev, "f32s")
    for size in (128, 256, 384):
        for seed in (20261, 1, 2, 3):
            x, y = calibration_pairs(dev, n=32, size=size, seed=seed)
            with torch.no_grad():
                s16 = ops.dists_score(*ops.dists_forward(x, y, p16, "f16"), a, b)
                s32 = ops.dists_score(*ops.dists_forward(x, y, p32, "f32s"), a, b)
            d = (s16 - s32).double()
            per_kind = [float(d[k::4].pow(2).mean().sqrt()) for k in range(4)]
            print(f"gain {gain} size {size} seed {seed}: max {float(d.abs().max()):.2e} rms {float(d.pow(2).mean().sqrt()):.2e} "
                  f"rms by kind (noise02, noise10, blur, indep) " + " ".join(f"{v:.2e}" for v in per_kind), flush=True)
