"""Does A-DISTS gain from running two batches on two HIP streams (the VALU-bound window pass of one beside the
MFMA-bound conv stack of the other)?  Development experiment, GPU box.
usage: python tools/gpu_adists_streams.py [B per stream]"""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd.ADISTS import ADISTS  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W = 1080, 1920
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.rand(B, 3, H, W, device=dev, generator=g) for _ in range(2)]
ys = [(x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1) for x in xs]
models = [ADISTS().to(dev).eval() for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run(n_streams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for k in range(steps):
            for s in range(n_streams):
                with torch.cuda.stream(streams[s]):
                    models[s](xs[s], ys[s], as_loss=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n_streams * steps * B / dt


for n in (1, 2):
    run(n, 3)
for rnd in range(3):
    for n in (1, 2):
        print(f"B={B} per stream, {n} stream(s): {run(n, 10):7.1f} pairs/s", flush=True)
