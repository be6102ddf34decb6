"""A few launches of the fused stage-1 kernel alone (B=8 1080p, f16), for rocprofv3 (tools/pmc_kernel.sh s1 conv1_pool -- python3 tools/gpu_s1_once.py)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
blob = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f16").to(dev)
g = torch.Generator(device=dev).manual_seed(1)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(b, 3, 1080, 1920, device=dev, generator=g)
y = (x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1)
for _ in range(4):
    ops.conv1_pool_stats(x, y, blob, "f16")
    ops.conv1_fused(torch.cat([x, y]), blob, "f16")  # the unfused stage 1 beside it
torch.cuda.synchronize()
print("done")
