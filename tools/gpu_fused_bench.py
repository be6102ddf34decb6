"""Time the fused stage-1 kernel alone (development aid; NQA_LIB selects an ablation build)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
dev = torch.device("cuda:0")
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f16").to(dev)
for (H, W, N) in ((256, 256, 64), (1080, 1920, 8)):
    x = torch.rand(N, 3, H, W, device=dev)
    ws = ops.Workspace()
    for _ in range(3):
        ops.vgg_pyramid(x[:2], packed, "f16", ws)
    import ctypes as C
    from nerf_qa_amd._lib import lib, ptr, stream_ptr, check
    out = torch.empty(N, H, W, 64, dtype=torch.float16, device=dev)
    # stage 1 only: through the debug entry (conv1_fused is reached via nqa_conv1_fused)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
    e1.record()
    torch.cuda.synchronize()
    print(f"{H}x{W} N={N}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
    ops.set_conv_variant(1 | 4)  # the tile form of stage 1
    out2 = torch.empty_like(out)
    for _ in range(3):
        check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out2), stream_ptr(dev)))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out2), stream_ptr(dev)))
    e1.record()
    torch.cuda.synchronize()
    print(f"   tile form: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us, max |diff| vs persistent form {(out2.float() - out.float()).abs().max().item():.3e}")
    ops.set_conv_variant(1)
    # the unfused alternative: conv1_1 (f16 NHWC out) + conv1_2 on the implicit-GEMM kernel
    a = ops.conv1_1(x, packed, "f16")
    for fn, name in ((lambda: ops.conv1_1(x, packed, "f16"), "conv1_1"), (lambda: ops.conv3x3_relu(a, 1, packed, "f16"), "conv1_2 igemm")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"   {name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
