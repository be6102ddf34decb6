"""Would a guard on the frames' HIGH-FREQUENCY energy separate the pairs on which the 16-bit rungs fail below 0.9 Mpx?
For the calibration pairs of one size class: per pair hf = min over (x, y) of the variance of frame - 5x5 box blur (mean
over the colour planes) next to |rung - f32s|; then, per threshold, the largest deviation among the pairs the guard would
NOT send to f32s and the share it would send (development aid for the next round, GPU box).
usage: python tools/gpu_hf_guard_study.py [class] [gain]"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS_pt as dp  # noqa: E402

cls = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device("cuda:0")
m = DISTS(vgg16_path="synth:1234" if gain == 1.0 else f"synth:1234:{gain}").to(dev).eval()
ab = np.load(dp._DATA)
a, b = torch.from_numpy(ab["alpha"]).to(dev), torch.from_numpy(ab["beta"]).to(dev)
modes = ("f16", "f16w", "f32m")
ws = ops.Workspace()
fam = ("noise02", "noise10", "blur", "indep", "noise10b", "white_bg", "black_bg", "smooth_fl")
hf, kinds, dev_of = [], [], {k: [] for k in modes}


def hfe(t):
    return (t - F.avg_pool2d(t, 5, 1, 2, count_include_pad=False)).var(dim=(2, 3)).mean(dim=1)


for n, ch, cw, seed in dp.AUTO_CLASSES[cls][1]:
    bs = max(1, min(n, (64 * 128 * 128) // (ch * cw) * 4))
    for i0 in range(0, n, bs):
        x, y = dp.calibration_pairs(dev, n=min(bs, n - i0), size=ch, seed=seed + 1000 * (i0 // bs), width=cw)
        sc = {}
        for prec in modes + ("f32s",):
            s1, s2 = ops.dists_forward(x, y, m._packed_weights(dev, prec), prec, ws)
            sc[prec] = ops.dists_score(s1, s2, a, b)
        for k in modes:
            dev_of[k].append((sc[k] - sc["f32s"]).abs().double().cpu().numpy())
        hf.append(torch.minimum(hfe(x), hfe(y)).cpu().numpy())
        kinds += [i % 8 for i in range(x.shape[0])]
hf, kinds = np.concatenate(hf), np.array(kinds)
d = {k: np.concatenate(v) for k, v in dev_of.items()}
print(f"class {cls} gain {gain}: {len(hf)} pairs; hf energy by family (median): " + ", ".join(f"{fam[f]} {np.median(hf[kinds == f]):.1e}" for f in range(8)))
for thr in (0.0, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3):
    keep = hf >= thr
    line = f"  guard at hf < {thr:.0e}: {100 * (1 - keep.mean()):5.1f} % of the pairs to f32s; of the rest, max / rms of |rung - f32s|:"
    for k in modes:
        v = d[k][keep]
        mx, rms = v.max(), np.sqrt((v * v).mean())
        line += f"  {k} {mx:.1e} / {rms:.1e} ({'admitted' if dp.admitted(mx, rms) else 'refused'})"
    print(line, flush=True)
