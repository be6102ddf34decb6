"""Quick per-kernel-class timing of the DISTS path (development aid, GPU box only)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

FLOP_PER_PIXEL = 2 * 9 * sum(ci * co / (4 ** s) for (ci, co, s) in
                             zip(ops.CONV_CIN, ops.CONV_COUT, ops.CONV_STAGE))  # approx (even sizes)


def run(prec, B, H, W, iters=5):
    dev = torch.device("cuda:0")
    packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), prec).to(dev)
    x = torch.rand(B, 3, H, W, device=dev)
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    ws = ops.Workspace()
    for _ in range(2):
        ops.dists_forward(x, y, packed, prec, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.dists_forward(x, y, packed, prec, ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    ops.timing_enable(True)
    ops.dists_forward(x, y, packed, prec, ws)
    t = ops.timing_collect()
    ops.timing_enable(False)
    flops = FLOP_PER_PIXEL * H * W * 2 * B
    print(f"{prec} B={B} {H}x{W}: {dt*1e3:.2f} ms/step  {B/dt:.1f} pairs/s  conv-stack {flops/dt/1e12:.1f} TFLOP/s eff")
    for k, (n, ms) in t.items():
        if n:
            print(f"    {k:10s} launches={n:3d} total={ms:.3f} ms")
    igemm_flops = flops - 2 * 27 * 64 * H * W * 2 * B
    print(f"    igemm-only rate: {igemm_flops / (t['conv_igemm'][1] * 1e-3) / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    for v in (0, 1):
        print("== conv variant", v)
        ops.set_conv_variant(v)
        run("f16", 32, 256, 256)
        run("f16", 4, 1080, 1920, iters=3)
    run("bf16", 32, 256, 256)
    run("f32", 32, 256, 256)
