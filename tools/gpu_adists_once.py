"""One A-DISTS forward at 1080p B=4 in f32s (for rocprofv3 --pmc over the window / chain kernels)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f32s").to(dev)
x = torch.rand(B, 3, 1080, 1920, device=dev)
y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
ws = ops.Workspace()
for _ in range(2):
    ops.adists_forward(x, y, packed, "f32s", ws)
torch.cuda.synchronize()
