"""What DISTS' `auto` precision decides for the three pinned stand-in weight sets, per frame-size class, and what the
one-time calibration of each class costs (run on the GPU box: `python tools/gpu_cal_check.py`)."""
import sys
import time

import torch

sys.path.insert(0, '/root/repo')
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

dev = torch.device("cuda:0")
for gain in (1.0, 1.3, 1.6):
    m = DISTS(vgg16_path=f"synth:1234:{gain}").to(dev).eval()
    for h, w in ((128, 128), (256, 256), (640, 960), (1080, 1920)):
        t0 = time.time()
        r = m.calibrate(dev, h, w)
        torch.cuda.synchronize()
        rungs = {k: {a: (round(b, 8) if isinstance(b, float) else b) for a, b in r[k].items()}
                 for k in ("f16", "f16w", "f32m4", "f32m", "f32m2")}
        print("gain", gain, f"{h}x{w} class", r["size_class"], r["sizes"], f"({time.time() - t0:.1f} s)", "->", r["choice"],
              rungs, flush=True)
