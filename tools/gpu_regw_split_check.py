"""conv2_1 in f32s: the register-weights kernel (conv3x3_regw_split_kernel) against the implicit GEMM it replaces
(conv variant bit 16 = first forms) and against a float64 convolution, on ragged and regular shapes; then the time of
both (development aid, GPU box).  usage: python tools/gpu_regw_split_check.py"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
convs = synth.vgg16_weights(1234)
packed = ops.pack_vgg_weights(convs, "f32s").to(dev)
w, b = torch.from_numpy(convs[2][0]).double(), torch.from_numpy(convs[2][1]).double()
g = torch.Generator().manual_seed(3)
for n, h, wd in ((3, 42, 33), (2, 4, 32), (1, 2, 16), (5, 21, 50), (2, 128, 128), (1, 67, 129), (16, 540, 960)):
    a = (torch.rand(n, h, wd, 64, generator=g) * 4 - 1).clamp_min(0)
    a[..., 5] = 0  # a dead channel
    enc = ops.split16_encode(a.to(dev))
    ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
    new = ops.split16_decode(ops.conv3x3_relu(enc, 2, packed, "f32s")).cpu()
    ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT | 16)
    old = ops.split16_decode(ops.conv3x3_relu(enc, 2, packed, "f32s")).cpu()
    ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
    if n * h * wd <= 200000:
        a_eff = ops.split16_decode(enc).cpu().double()  # what both kernels actually see
        ref = F.relu(F.conv2d(a_eff.permute(0, 3, 1, 2), w, b, padding=1)).permute(0, 2, 3, 1)
        sc = ref.abs().max().item()
        print(f"{n}x{h}x{wd}: new vs f64 {(new.double() - ref).abs().max().item() / sc:.2e}  old vs f64 {(old.double() - ref).abs().max().item() / sc:.2e}  "
              f"new vs old {(new - old).abs().max().item() / sc:.2e}", flush=True)
    else:
        print(f"{n}x{h}x{wd}: new vs old {(new - old).abs().max().item() / old.abs().max().item():.2e}", flush=True)
# repeats: the same launch 40 times on ragged maps whose blocks own two tiles each (how a DMA lane that addressed LDS behind
# its slot was found: single wrong pixels in about half of the launches)
for n, h, wd in ((2, 97, 131), (4, 49, 66)):
    a = ops.split16_encode(((torch.rand(n, h, wd, 64, generator=g) * 4 - 1).clamp_min(0)).to(dev))
    first = ops.conv3x3_relu(a, 2, packed, "f32s").clone()
    same = sum(bool(torch.equal(first.view(torch.int32), ops.conv3x3_relu(a, 2, packed, "f32s").view(torch.int32))) for _ in range(40))
    print(f"{n}x{h}x{wd}: {same} of 40 repeats bit-equal", flush=True)
a = ops.split16_encode((torch.rand(16, 540, 960, 64, generator=g)).to(dev))
for name, v in (("register weights", ops.DEFAULT_CONV_VARIANT), ("implicit GEMM", ops.DEFAULT_CONV_VARIANT | 16)):
    ops.set_conv_variant(v)
    for _ in range(3):
        ops.conv3x3_relu(a, 2, packed, "f32s")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3_relu(a, 2, packed, "f32s")
    e1.record()
    torch.cuda.synchronize()
    print(f"conv2_1 f32s, 16 images of 540x960, {name}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
ops.set_conv_variant(ops.DEFAULT_CONV_VARIANT)
