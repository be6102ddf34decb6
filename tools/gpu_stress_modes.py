"""Random frame sizes through DISTS in every precision mode against the CPU oracle, per stand-in weight gain
(development aid, GPU box): the error distribution behind the `auto` calibration's budgets.
usage: python tools/gpu_stress_modes.py [N] [lo] [hi] [gains...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import synth  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from oracle import dists_oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
LO = int(sys.argv[2]) if len(sys.argv) > 2 else 96
HI = int(sys.argv[3]) if len(sys.argv) > 3 else 300
GAINS = [float(g) for g in sys.argv[4:]] or [1.0, 1.3, 1.6]
dev = torch.device("cuda:0")
torch.set_num_threads(16)
for gain in GAINS:
    spec = f"synth:1234:{gain}"
    convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234, gain))
    models = {p: DISTS(precision=p, vgg16_path=spec).to(dev).eval() for p in ("f16", "f16w", "f32m4", "f32m", "f32m2", "f32s")}
    auto = DISTS(vgg16_path=spec).to(dev).eval()
    choice = auto.calibrate(dev, LO, LO)["choice"]  # (the class of the smallest frames drawn)
    a, b = auto.alpha.detach().cpu(), auto.beta.detach().cpu()
    rng = np.random.default_rng(4242)
    errs = {p: [] for p in models}
    for i in range(N):
        h, w, bsz = int(rng.integers(LO, HI + 1)), int(rng.integers(LO, HI + 1)), int(rng.integers(1, 5))
        kinds = [synth.KINDS[int(k)] for k in rng.integers(0, 4, bsz)]
        seeds = [int(s) for s in rng.integers(0, 10 ** 6, bsz)]
        xn, yn = synth.frame_batch(seeds, h, w, kinds)
        x, y = torch.from_numpy(xn), torch.from_numpy(yn)
        with torch.no_grad():
            ref = dists_oracle.dists(x, y, convs, a, b)
            for p, m in models.items():
                errs[p] += (m(x.to(dev), y.to(dev)).cpu() - ref).abs().tolist()
    line = f"gain {gain} (auto -> {choice}), {sum(len(v) for v in errs.values()) // 6} pairs of {LO}..{HI} px:"
    for p, v in errs.items():
        v = np.array(v)
        line += f"  {p}: max {v.max():.2e} p99 {np.quantile(v, 0.99):.2e} rms {np.sqrt((v * v).mean()):.2e}"
    print(line, flush=True)
    assert np.max(errs[choice]) <= 1e-4, (gain, choice)
