"""pool+statistics pass time per step (library event ring), DISTS B=8 1080p and B=32 256x256, f16 and f32s.
Run once per library build (NQA_LIB=...) on the same box for A/B comparisons."""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

dev = torch.device("cuda:0")
print("lib:", os.environ.get("NQA_LIB", "default"))
for (b, h, w) in ((8, 1080, 1920), (32, 256, 256)):
    for prec in ("f16", "f32s"):
        m = DISTS(precision=prec, vgg16_path="synth:1234").to(dev).eval()
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.rand(b, 3, h, w, device=dev, generator=g)
        y = (x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1)
        with torch.no_grad():
            for _ in range(3):
                m(x, y)
            torch.cuda.synchronize()
            ops.timing_enable(True)
            for _ in range(10):
                m(x, y)
            torch.cuda.synchronize()
        kt = ops.timing_collect()
        ops.timing_enable(False)
        esz = 2 if prec == "f16" else 4
        alg = sum(2 * b * (hh * ww * c * esz + ((hh + 1) // 2) * ((ww + 1) // 2) * c * esz)
                  for (hh, ww), c in zip(ops.pyramid_dims(h, w)[:4], ops.CHNS[1:5]))
        ms = kt["l2pool"][1] / 10
        print(f"  {h}x{w} B={b} {prec}: pool+stats {ms:.4f} ms/step = {alg / ms / 1e6:.0f} GB/s; conv {kt['conv_igemm'][1] / 10:.3f} ms/step",
              flush=True)
        del m, x, y
        torch.cuda.empty_cache()
