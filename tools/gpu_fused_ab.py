"""Same-process A/B of the fused taps (stage 1 and conv2_2 with their L2-pool + statistics in one kernel each,
nqa_conv1_pool.hip / nqa_conv_pool.hip) against the unfused kernels (nqa_set_conv_variant + 128 / + 64), GPU box: DISTS
B=8 1080p (or `--size H W --batch B`), per-class kernel times from the library's event ring and the step time, the forms
taken in turn so that clock drift hits all of them.
usage: python tools/gpu_fused_ab.py [--prec f16] [--size 1080 1920] [--batch 8] [--rounds 4]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="f16")
ap.add_argument("--size", type=int, nargs=2, default=(1080, 1920))
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = DISTS(precision=a.prec, vgg16_path="synth:1234").to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(a.batch, 3, *a.size, device=dev, generator=g)
y = (x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1)
FORMS = ((1 + 64 + 128, "all unfused        "), (1 + 128, "tap 2 fused        "), (1, "stage 1 + tap 2 fused"))
res = {v: [] for v, _ in FORMS}
with torch.no_grad():
    for v, _ in FORMS:
        ops.set_conv_variant(v)
        for _ in range(3):
            m(x, y)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for v, _ in (FORMS if r % 2 == 0 else FORMS[::-1]):
            ops.set_conv_variant(v)
            ops.timing_enable(True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                m(x, y)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps * 1e3
            kt = ops.timing_collect()
            ops.timing_enable(False)
            res[v].append((dt, kt["conv_igemm"][1] / a.steps, kt["l2pool"][1] / a.steps, kt["stats"][1] / a.steps))
ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
for v, name in FORMS:
    for dt, c, p, s in res[v]:
        print(f"{name}: step {dt:7.3f} ms  conv class {c:7.3f}  pool class {p:6.3f}  stats {s:5.3f}")
med = lambda v, i: sorted(t[i] for t in res[v])[len(res[v]) // 2]  # noqa: E731
base = FORMS[0][0]
for v, name in FORMS:
    print(f"median {name}: step {med(v, 0):.3f} ms ({(med(base, 0) / med(v, 0) - 1) * 100:+.2f} % vs all unfused, "
          f"{a.batch / med(v, 0) * 1e3:.1f} pairs/s); conv class {med(v, 1):.3f}; pool class {med(v, 2):.3f}")
