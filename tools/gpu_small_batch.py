"""Throughput vs batch size through the module surface (is the small-batch path launch-bound?)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import sys, time, warnings
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from nerf_qa_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = DISTS().to(dev).eval()
for (H, W) in ((256, 256), (1080, 1920)):
    for B in (1, 2, 4, 8, 32) if H == 256 else (1, 2, 8):
        x = torch.rand(B, 3, H, W, device=dev)
        y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
        with torch.no_grad():
            for _ in range(5):
                m(x, y)
            torch.cuda.synchronize()
            n = 50 if H == 256 else 10
            t0 = time.perf_counter()
            for _ in range(n):
                s = m(x, y)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            ops.timing_enable(True)
            m(x, y)
            t = ops.timing_collect()
            ops.timing_enable(False)
        gpu = sum(v[1] for v in t.values())
        print(f"{H}x{W} B={B}: {dt * 1e3:.3f} ms/call wall, {gpu:.3f} ms of kernels, {B / dt:.0f} pairs/s", flush=True)
