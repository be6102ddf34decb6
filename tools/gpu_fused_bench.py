"""Time the stage-1 kernels alone (development aid; NQA_LIB selects an ablation build): the shipped register-weights
form (variant 1), the first persistent two-phase form (1+16) and the tile form (1+4); outputs compared."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
from nerf_qa_amd._lib import lib, ptr, stream_ptr, check  # noqa: E402
dev = torch.device("cuda:0")
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f16").to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (H, W, N) in ((256, 256, 64), (1080, 1920, 16)):
    x = torch.rand(N, 3, H, W, device=dev)
    flops = 2 * 9 * 64 * 64 * H * W * N
    ref = None
    for variant, name in ((1, "register weights"), (1 + 16, "two-phase"), (1 + 4, "tile")):
        ops.set_conv_variant(variant)
        out = torch.empty(N, H, W, 64, dtype=torch.float16, device=dev)
        best = 1e9
        for rnd in range(3):
            for _ in range(2):
                check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        d = 0.0 if ref is None else (out.float() - ref.float()).abs().max().item()
        ref = out if ref is None else ref
        print(f"{H}x{W} N={N} {name:18s}: {best * 1e3:8.1f} us  {flops / best / 1e9:6.0f} TF/s (conv1_2 FLOPs)  max|diff| {d:.2e}", flush=True)
    ops.set_conv_variant(1)
