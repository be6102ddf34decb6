"""Run one conv layer repeatedly (for rocprofv3 --pmc).  usage: gpu_one_layer.py LAYER VARIANT [256|1080] [REPS]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

layer, variant = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "256"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
H, W, N = (256, 256, 64) if which == "256" else (1080, 1920, 8)
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f16").to(dev)
h, w = ops.pyramid_dims(H, W)[ops.CONV_STAGE[layer]]
a = (torch.rand(N, h, w, ops.CONV_CIN[layer], device=dev) - 0.5).clamp_min(0).half()
ops.set_conv_variant(variant)
for _ in range(reps):
    ops.conv3x3_relu(a, layer, packed, "f16")
torch.cuda.synchronize()
