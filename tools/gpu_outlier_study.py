"""Where do the f16-class modes' OUTLIER pairs come from?  Per tap and per term (S1 / S2) decomposition of
score(mode) - score(f32s) on the calibration pairs (development aid, GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch.DISTS_pt import _DATA, AUTO_CAL_SETS, calibration_pairs  # noqa: E402

dev = torch.device("cuda:0")
gain = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
modes = sys.argv[2:] or ["f16", "f16w"]
ab = np.load(_DATA)
a, b = torch.from_numpy(ab["alpha"]).reshape(-1).to(dev).double(), torch.from_numpy(ab["beta"]).reshape(-1).to(dev).double()
w = a.sum() + b.sum()
offs = [0, 3, 67, 195, 451, 963, 1475]
m = DISTS(vgg16_path=f"synth:1234:{gain}", precision="f32s").to(dev).eval()
for mode in modes:
    rows = []
    for n, h, wd, seed in AUTO_CAL_SETS:
        x, y = calibration_pairs(dev, n=n, size=h, seed=seed, width=wd)
        with torch.no_grad():
            r1, r2 = ops.dists_forward(x, y, m._packed_weights(dev, "f32s"), "f32s")
            s1, s2 = ops.dists_forward(x, y, m._packed_weights(dev, mode), mode)
        d1 = -(a / w) * (s1 - r1).double()   # per (pair, channel) contribution to the score difference
        d2 = -(b / w) * (s2 - r2).double()
        per = torch.stack([torch.stack([d1[:, offs[k]:offs[k + 1]].sum(1), d2[:, offs[k]:offs[k + 1]].sum(1)], 1)
                           for k in range(6)], 1)  # (pairs, tap, term)
        # largest single-channel contribution of each pair and where it sits
        both = torch.cat([d1, d2], 1).abs()
        top, idx = both.max(1)
        rows.append((per, top, idx, (s2 - r2).abs().max(1)[0]))
    per = torch.cat([r[0] for r in rows])
    tot = per.sum((1, 2))
    print(f"== gain {gain} mode {mode}: total |d| max {tot.abs().max():.2e} rms {tot.pow(2).mean().sqrt():.2e}")
    rms = per.pow(2).mean(0).sqrt().cpu().numpy()
    print("   rms contribution by tap (rows) and term (S1, S2):")
    for k in range(6):
        print(f"     tap {k}: S1 {rms[k, 0]:.2e}  S2 {rms[k, 1]:.2e}")
    worst = tot.abs().argsort(descending=True)[:6]
    top = torch.cat([r[1] for r in rows]); idx = torch.cat([r[2] for r in rows]); ds2 = torch.cat([r[3] for r in rows])
    for i in worst.tolist():
        c = int(idx[i]); term = "S1" if c < 1475 else "S2"; c %= 1475
        tap = max(k for k in range(6) if offs[k] <= c)
        print(f"   pair {i} (kind {i % 4}): d {tot[i]:+.2e}; by tap S2 " + " ".join(f"{per[i, k, 1]:+.1e}" for k in range(6))
              + f"; largest single channel {float(top[i]):.1e} ({term}, tap {tap}); max|dS2| {float(ds2[i]):.1e}")
