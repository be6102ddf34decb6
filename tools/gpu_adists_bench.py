"""A-DISTS timing by kernel class (development aid, GPU box only)."""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
dev = torch.device("cuda:0")
PREC = sys.argv[1] if len(sys.argv) > 1 else "f32s"
VARIANT = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # +8: the per-wave-loads form of the window pass
ops.set_conv_variant(VARIANT)
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), PREC).to(dev)
for (B, H, W) in ((32, 256, 256), (8, 1080, 1920)):
    x = torch.rand(B, 3, H, W, device=dev)
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    ws = ops.Workspace()
    for _ in range(2):
        ops.adists_forward(x, y, packed, PREC, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ops.adists_forward(x, y, packed, PREC, ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ops.timing_enable(True)
    ops.adists_forward(x, y, packed, PREC, ws)
    t = ops.timing_collect()
    ops.timing_enable(False)
    print(f"variant {VARIANT} A-DISTS B={B} {H}x{W}: {dt*1e3:.2f} ms/step {B/dt:.1f} pairs/s", {k: (v[0], round(v[1], 3)) for k, v in t.items() if v[0]}, flush=True)
