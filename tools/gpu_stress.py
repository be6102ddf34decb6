"""One-off randomized stress: many random frame sizes through both metrics against the CPU oracle, with a fenced
workspace (prints the worst deviations; exits non-zero on a parity or fence failure)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import json, sys, warnings
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from nerf_qa_amd import ops, synth  # noqa: E402
from nerf_qa_amd.ADISTS import ADISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from oracle import adists_oracle, dists_oracle  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
HI = int(sys.argv[2]) if len(sys.argv) > 2 else 160
LO = int(sys.argv[3]) if len(sys.argv) > 3 else 1
BMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
torch.set_num_threads(16)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m16, m32s, a = DISTS(precision="f16").to(dev).eval(), DISTS(precision="f32s").to(dev).eval(), ADISTS().to(dev).eval()
convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234))
rng = np.random.default_rng(99)
worst = {"f16": 0.0, "f32s": 0.0, "a32s": 0.0}
flips = 0
cases = []  # inputs of every A-DISTS deviation above 1e-4, for oracle/knife_edge_study.py
for i in range(N):
    h, w, b = int(rng.integers(LO, HI + 1)), int(rng.integers(LO, HI + 1)), int(rng.integers(1, BMAX + 1))
    kinds = [synth.KINDS[int(k)] for k in rng.integers(0, 4, b)]
    seeds = [int(s) for s in rng.integers(0, 10 ** 6, b)]
    xn, yn = synth.frame_batch(seeds, h, w, kinds)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    with torch.no_grad():
        ref = dists_oracle.dists(x, y, convs, m16.alpha.detach().cpu(), m16.beta.detach().cpu())
        aref = adists_oracle.adists(x, y, convs)
        xd, yd = x.to(dev), y.to(dev)
        e16 = (m16(xd, yd).cpu() - ref).abs().max().item()
        e32 = (m32s(xd, yd).cpu() - ref).abs().max().item()
        ga = a(xd, yd, as_loss=False).cpu()
    ok = ~torch.isnan(aref)
    assert torch.equal(torch.isnan(ga), torch.isnan(aref)), (h, w, b, ga, aref)
    ea = (ga[ok] - aref[ok]).abs().max().item() if ok.any() else 0.0
    if ea > 1e-4:
        flips += 1
        print(f"  A-DISTS knife edge? {h}x{w} b={b} kinds={kinds}: |d|={ea:.2e}", flush=True)
        for j in range(b):
            if ok[j] and abs(ga[j].item() - aref[j].item()) > 1e-4:
                cases.append({"iter": i, "h": h, "w": w, "seed": seeds[j], "kind": kinds[j], "slot": j, "batch": b,
                              "hip_f32s": ga[j].item(), "oracle_f32": aref[j].item()})
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(cases, open("gpurun_out/knife_cases.json", "w"), indent=1)
    # forced f16 is held to the bar only where DISTS' `auto` default selects it (>= 128 x 128 pixels); below that
    # its tail is the reason `auto` switches to f32s, and the value is just recorded
    assert (e16 <= 1e-4 or h * w < 128 * 128) and e32 <= 5e-6, (h, w, b, e16, e32)
    worst = {"f16": max(worst["f16"], e16), "f32s": max(worst["f32s"], e32), "a32s": max(worst["a32s"], ea)}
    if i % 25 == 24 or HI > 400:  # (large frames: a line per case, minutes apart)
        print(i + 1, (h, w, b), worst, flush=True)
print("done", N, worst, "A-DISTS cases above 1e-4:", flips)
