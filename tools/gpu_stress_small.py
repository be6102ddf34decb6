"""Tail of the f16 DISTS error on small frames (where the statistics run over few pixels)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import sys, warnings
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import synth  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from oracle import dists_oracle  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
HI = int(sys.argv[2]) if len(sys.argv) > 2 else 64
LO = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
torch.set_num_threads(16)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m16 = DISTS(precision="f16").to(dev).eval()
convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234))
rng = np.random.default_rng(7)
errs = []
for i in range(N):
    h, w, b = int(rng.integers(LO, HI + 1)), int(rng.integers(LO, HI + 1)), 2
    kinds = [synth.KINDS[int(k)] for k in rng.integers(0, 4, b)]
    xn, yn = synth.frame_batch([int(s) for s in rng.integers(0, 10 ** 6, b)], h, w, kinds)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    with torch.no_grad():
        ref = dists_oracle.dists(x, y, convs, m16.alpha.detach().cpu(), m16.beta.detach().cpu())
        e = (m16(x.to(dev), y.to(dev)).cpu() - ref).abs().max().item()
    errs.append((e, h, w))
errs.sort(reverse=True)
print("worst 8:", [(f"{e:.2e}", h, w) for e, h, w in errs[:8]])
a = np.array([e for e, _, _ in errs])
print("N", N, "max", a.max(), "p99", np.quantile(a, 0.99), "median", np.median(a), "count > 1e-4:", int((a > 1e-4).sum()))
