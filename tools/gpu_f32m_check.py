"""First light of the mixed mode (development aid, GPU box): taps and scores of f32m against the oracle / f32s, the
calibration of the three pinned weight sets, and step times of f16 / f32m / f32s at 1080p B=8 and 256x256 B=32."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from oracle import dists_oracle  # noqa: E402

dev = torch.device("cuda:0")
convs = synth.vgg16_weights(1234)
oc = dists_oracle.convs_from_numpy(convs)
x, _ = synth.frame_batch([5, 6], 97, 131)
x = torch.from_numpy(x)
ref = dists_oracle.vgg_pyramid(x, oc)[1:]
for prec in ("f16", "f16w", "f32m4", "f32m", "f32m2", "f32s"):
    packed = ops.pack_vgg_weights(convs, prec).to(dev)
    taps = ops.vgg_pyramid(x.to(dev), packed, prec)
    errs = []
    for k, (t, r) in enumerate(zip(taps, ref)):
        got = ops.nhwc_to_nchw_f32(t, ops.tap_prec(prec, k)).cpu()
        errs.append((got - r).abs().max().item() / r.abs().max().item())
    print(prec, "tap rel err", " ".join(f"{e:.1e}" for e in errs), flush=True)

for gain in (1.0, 1.3, 1.6):
    m = DISTS(vgg16_path=f"synth:1234:{gain}").to(dev).eval()
    print("gain", gain, m.calibrate(dev, 256, 256), flush=True)

for (b, h, w) in ((8, 1080, 1920), (32, 256, 256)):
    g = torch.Generator(device=dev).manual_seed(1)
    xx = torch.rand(b, 3, h, w, device=dev, generator=g)
    yy = (xx + 0.1 * torch.randn(xx.shape, device=dev, generator=g)).clamp_(0, 1)
    base = None
    for prec in ("f32s", "f32m2", "f32m", "f32m4", "f16w", "f16"):
        m = DISTS(precision=prec, vgg16_path="synth:1234").to(dev).eval()
        with torch.no_grad():
            for _ in range(3):
                s = m(xx, yy)
            torch.cuda.synchronize()
            ops.timing_enable(True)
            t0 = time.perf_counter()
            for _ in range(10):
                s = m(xx, yy)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
        kt = ops.timing_collect()
        ops.timing_enable(False)
        if base is None:
            base = s
        print(f"{h}x{w} B={b} {prec}: {dt * 1e3:.2f} ms/step = {b / dt:.1f} pairs/s; max|d vs f32s| = {(s - base).abs().max().item():.2e}; "
              + " ".join(f"{k}={v[1] / 10:.2f}" for k, v in kt.items() if v[0]), flush=True)
        del m
        torch.cuda.empty_cache()
