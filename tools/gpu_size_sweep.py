"""DISTS / A-DISTS throughput across frame sizes (module surface, default precisions)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import sys, time, warnings
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd.ADISTS import ADISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    d, a = DISTS().to(dev).eval(), ADISTS().to(dev).eval()
    d16 = DISTS(precision="f16").to(dev).eval()
for (H, W, B) in ((256, 256, 32), (512, 512, 16), (800, 800, 8), (720, 1280, 8), (1080, 1920, 8), (2160, 3840, 2)):
    x = torch.rand(B, 3, H, W, device=dev)
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    row = f"{H}x{W} B={B}:"
    for name, fn in ((f"DISTS auto->{d.precision_for(H, W, dev)}", lambda: d(x, y)), ("DISTS f16", lambda: d16(x, y)),
                     ("A-DISTS", lambda: a(x, y, as_loss=False))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            n = 20 if H <= 512 else 6
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        row += f"  {name} {B / dt:9.1f} pairs/s ({dt * 1e3:8.2f} ms)"
    print(row, flush=True)
