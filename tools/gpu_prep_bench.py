"""Timing of the input-preparation kernels against their HBM floor (development aid, GPU box only)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, prep  # noqa: E402
dev = torch.device("cuda:0")
N, H, W = 64, 1080, 1920
f = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device=dev)


def bench(name, fn, bytes_moved, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms:8.3f} ms  {N / ms * 1e3:10.0f} frames/s  {bytes_moved / ms / 1e6:8.1f} GB/s algorithmic", flush=True)


src = N * H * W * 3
ws = ops.Workspace()
bench("ToTensor 1080p", lambda: ops.u8hwc_to_f32nchw(f), src + src * 4)
bench("interp256 (fused u8->f32 256x256)", lambda: prep.prepare_frames(f, "interp256"), N * 256 * 256 * 3 * (4 + 4))
bench("pil256 (1080p->256x256)", lambda: prep.prepare_frames(f, "pil256", ws=ws), src + N * H * 256 * 3 * 2 + N * 256 * 256 * 3 * 5)
bench("pil256 keep aspect (256x455)", lambda: prep.prepare_frames(f, "pil256", keep_aspect_ratio=True, ws=ws),
      src + N * H * 455 * 3 * 2 + N * 256 * 455 * 3 * 5)
bench("equal_pixels (192x341)", lambda: prep.prepare_frames(f, "equal_pixels"), N * 192 * 341 * 3 * 8)
