"""Which calibration pairs carry each rung's deviation from f32s?  (development aid, GPU box)

Re-runs the pairs of one size class of DISTS' `auto` calibration and prints, per content family (pair index mod 8, see
calibration_pairs) and rung, the max and rms of score_rung - score_f32s, and per tap which similarity moved; the three
worst pairs of the named rung are saved to gpurun_out/cal_worst_<class>.npz for tools/cpu_prec_layers.py-style emulation.

usage: python tools/gpu_cal_pairs.py [class 0..3] [gain] [rung to save]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS_pt as dp  # noqa: E402

cls = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gain = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
save = sys.argv[3] if len(sys.argv) > 3 else "f32m2"
dev = torch.device("cuda:0")
m = DISTS(precision="auto", vgg16_path=f"synth:1234:{gain}" if gain != 1.0 else "synth:1234").to(dev).eval()
ab = np.load(dp._DATA)
a, b = torch.from_numpy(ab["alpha"]).to(dev), torch.from_numpy(ab["beta"]).to(dev)
modes = dp.LADDER[:-1]
ws = ops.Workspace()
fam = ("noise02", "noise10", "blur", "indep", "noise10b", "white_bg", "black_bg", "smooth_floaters")
CH = (3, 64, 128, 256, 512, 512)
edges = np.cumsum((0,) + CH)
rows = {k: [] for k in modes}
sims_d = {k: [] for k in modes}
kinds, keep = [], []
for n, ch, cw, seed in dp.AUTO_CLASSES[cls][1]:
    bs = max(1, min(n, (64 * 128 * 128) // (ch * cw) * 4))
    for i0 in range(0, n, bs):
        x, y = dp.calibration_pairs(dev, n=min(bs, n - i0), size=ch, seed=seed + 1000 * (i0 // bs), width=cw)
        sc, sm = {}, {}
        for prec in modes + ("f32s",):
            s1, s2 = ops.dists_forward(x, y, m._packed_weights(dev, prec), prec, ws)
            sc[prec], sm[prec] = ops.dists_score(s1, s2, a, b), (s1, s2)
        for k in modes:
            rows[k].append((sc[k] - sc["f32s"]).double().cpu().numpy())
            # contribution of each tap's S1 / S2 movement to the score (alpha, beta weighted)
            d1 = ((sm[k][0] - sm["f32s"][0]) * a.view(1, -1)).double().cpu().numpy()
            d2 = ((sm[k][1] - sm["f32s"][1]) * b.view(1, -1)).double().cpu().numpy()
            sims_d[k].append(np.stack([[d1[:, edges[t]:edges[t + 1]].sum(1), d2[:, edges[t]:edges[t + 1]].sum(1)] for t in range(6)]))
        kinds += [i % 8 for i in range(x.shape[0])]
        keep.append((x.cpu().numpy(), y.cpu().numpy(), (ch, cw)))
kinds = np.array(kinds)
print(f"class {cls} gain {gain}: {len(kinds)} pairs")
for k in modes:
    d = np.concatenate(rows[k])
    line = f"  {k:6s} all: max {np.abs(d).max():.2e} rms {np.sqrt((d * d).mean()):.2e} |"
    for f in range(8):
        df = d[kinds == f]
        line += f" {fam[f]} {np.abs(df).max():.1e}/{np.sqrt((df * df).mean()):.1e}"
    print(line)
d = np.concatenate(rows[save])
worst = np.argsort(-np.abs(d))[:3]
sd = np.concatenate(sims_d[save], axis=2)  # (6 taps, 2, pairs)
for wi in worst:
    print(f"  worst of {save}: pair {wi} family {fam[kinds[wi]]} d={d[wi]:+.2e}; per tap (alpha*dS1 | beta*dS2) = "
          + "  ".join(f"t{t}: {sd[t, 0, wi]:+.1e}|{sd[t, 1, wi]:+.1e}" for t in range(6)))
    print("      every rung on this pair: " + "  ".join(f"{k} {np.concatenate(rows[k])[wi]:+.2e}" for k in modes))
# save the worst pairs (frames of the first calibration set only have the same size; locate each in its batch)
out = {}
off = 0
for xb, yb, sz in keep:
    for j, wi in enumerate(worst):
        if off <= wi < off + xb.shape[0]:
            out[f"x{j}"], out[f"y{j}"], out[f"d{j}"] = xb[wi - off], yb[wi - off], d[wi]
    off += xb.shape[0]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"cal_worst_{cls}.npz"), gain=gain, **out)
