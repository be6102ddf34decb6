"""Eager enqueue vs hipGraph replay of one DISTS forward (f16 auto) at 256 x 256 for several batch sizes, and at 1080p B=8:
how much of a step is launch gaps (development aid)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
dev = torch.device("cuda:0")
model = DISTS().to(dev).eval()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for (h, w, b, n) in ((256, 256, 1, 300), (256, 256, 2, 300), (256, 256, 8, 200), (256, 256, 32, 100), (1080, 1920, 8, 20)):
        x = torch.rand(b, 3, h, w, device=dev)
        y = (x + 0.05 * torch.randn_like(x)).clamp(0, 1)
        eager = lambda: model(x, y, batch_average=False)
        for _ in range(3):
            eager()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = eager()
        te = min(timed(eager, n) for _ in range(3))
        tg = min(timed(g.replay, n) for _ in range(3))
        print(f"{h}x{w} B={b}: eager {te:.3f} ms ({b / te * 1e3:.0f} pairs/s)   graph replay {tg:.3f} ms ({b / tg * 1e3:.0f} pairs/s)   {te / tg:.3f}x", flush=True)
