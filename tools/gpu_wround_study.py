"""Does error-feedback rounding of the f16 weights (residual carried along the contraction) shrink the f16 score error?"""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
from oracle import dists_oracle  # noqa: E402
dev = torch.device("cuda:0")
torch.set_num_threads(16)
np_convs = synth.vgg16_weights(1234)
convs = dists_oracle.convs_from_numpy(np_convs)
ab = np.load(__file__.rsplit("/", 2)[0] + "/nerf_qa_amd/data/dists_alpha_beta.npz")
alpha, beta = torch.from_numpy(ab["alpha"]), torch.from_numpy(ab["beta"])


def diffuse(w, order):
    """round to f16 carrying the residual to the next weight of the same output channel; order: 'tap' = taps of one
    input channel are neighbours, 'cin' = input channels of one tap are neighbours"""
    cout, cin = w.shape[:2]
    v = w.reshape(cout, cin, 9).astype(np.float64)
    if order == "cin":
        v = v.transpose(0, 2, 1)
    flat = v.reshape(cout, -1)
    out = np.empty_like(flat)
    r = np.zeros(cout)
    for i in range(flat.shape[1]):
        t = flat[:, i] + r
        q = t.astype(np.float16).astype(np.float64)
        out[:, i] = q
        r = t - q
    out = out.reshape(v.shape)
    if order == "cin":
        out = out.transpose(0, 2, 1)
    return out.reshape(w.shape).astype(np.float32)


packs = {"rne": ops.pack_vgg_weights(np_convs, "f16").to(dev)}
for order in ("tap", "cin"):
    mod = [(np_convs[0][0], np_convs[0][1])] + [(diffuse(w, order), b) for w, b in np_convs[1:]]
    packs[order] = ops.pack_vgg_weights(mod, "f16").to(dev)
rng = np.random.default_rng(7)
errs = {k: [] for k in packs}
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    lo, hi = (1, 64) if i % 2 == 0 else (64, 200)
    h, w = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
    xn, yn = synth.frame_batch([int(s) for s in rng.integers(0, 10 ** 6, 2)], h, w, [synth.KINDS[int(k)] for k in rng.integers(0, 4, 2)])
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    ref = dists_oracle.dists(x, y, convs, alpha.view(1, -1, 1, 1), beta.view(1, -1, 1, 1))
    for k, pk in packs.items():
        s1, s2 = ops.dists_forward(x.to(dev), y.to(dev), pk, "f16")
        got = ops.dists_score(s1, s2, alpha.to(dev), beta.to(dev)).cpu()
        errs[k].append((got - ref).abs().max().item())
for k, e in errs.items():
    e = np.array(e)
    print(f"{k:4s}: small frames max {e[0::2].max():.2e} p99 {np.quantile(e[0::2], .99):.2e} median {np.median(e[0::2]):.2e} | "
          f"64-200 px max {e[1::2].max():.2e} median {np.median(e[1::2]):.2e}")
