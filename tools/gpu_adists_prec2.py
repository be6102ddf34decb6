"""Which rounding hurts A-DISTS in f16: weights or activations?  (development aid)"""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from nerf_qa_amd import ops, synth  # noqa: E402
from test_gpu_fullsize import _frames  # noqa: E402
dev = torch.device("cuda:0")
convs = synth.vgg16_weights(1234)
r16 = [(torch.from_numpy(w).half().float().numpy(), b) for w, b in convs]
p32 = ops.pack_vgg_weights(convs, "f32").to(dev)
p32w16 = ops.pack_vgg_weights(r16, "f32").to(dev)
p16 = ops.pack_vgg_weights(convs, "f16").to(dev)
p32s = ops.pack_vgg_weights(convs, "f32s").to(dev)
for (b, h, w) in ((8, 256, 256), (4, 128, 160), (2, 1080, 1920)):
    x, y = _frames(b, h, w, dev, 11)
    ref = 1 - ops.adists_forward(x, y, p32, "f32")
    a = 1 - ops.adists_forward(x, y, p32w16, "f32")
    c = 1 - ops.adists_forward(x, y, p16, "f16")
    e = 1 - ops.adists_forward(x, y, p32s, "f32s")
    print(f"{h}x{w}: split-f16 (f32s): max|d|={(e - ref).abs().max().item():.2e}")
    print(f"{h}x{w}: f32 kernels + f16-rounded weights: max|d|={(a - ref).abs().max().item():.2e} ; full f16: {(c - ref).abs().max().item():.2e}")
    print("   per pair (w16):", [f"{v:.1e}" for v in (a - ref).tolist()], " (f16):", [f"{v:.1e}" for v in (c - ref).tolist()])
