"""Per-layer timing of the implicit-GEMM conv (development aid, GPU box only).

usage: python tools/gpu_layer_bench.py [256|1080]   (NQA_LIB selects an ablation build)
Variants are interleaved per layer in one process after a global warm-up (DVFS).
"""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "256"
H, W, N = (256, 256, 64) if which == "256" else (1080, 1920, 16)
prec = os.environ.get("NQA_TOOL_PREC", "f16")
DT = {"f16": torch.float16, "bf16": torch.bfloat16}.get(prec, torch.float32)
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), prec).to(dev)
dims = ops.pyramid_dims(H, W)
VARIANTS = tuple(int(v) for v in os.environ.get('NQA_TOOL_VARIANTS', '0,1,2').split(','))


def time_layer(a, layer, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv3x3_relu(a, layer, packed, prec)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# warm-up: ~0.5 s of conv work so clocks settle
a = (torch.rand(N, dims[2][0], dims[2][1], 256, device=dev) - 0.5).clamp_min(0).to(DT)
time_layer(a, 5, 300 if which == "256" else 40)
res = {v: [] for v in VARIANTS}
tot = {v: [0.0, 0.0] for v in VARIANTS}
for layer in range(1, 13):
    h, w = dims[ops.CONV_STAGE[layer]]
    cin, cout = ops.CONV_CIN[layer], ops.CONV_COUT[layer]
    a = (torch.rand(N, h, w, cin, device=dev) - 0.5).clamp_min(0).to(DT)
    if prec == "f32s":
        a = ops.split16_encode(a.contiguous())  # conv layers read split16 records in this mode
    if os.environ.get('NQA_TOOL_ZERO'):
        a.zero_()  # clock check: zero operands toggle nothing, so the chip holds its clock (DVFS)
    fl = 2 * 9 * cin * cout * h * w * N
    best = {v: 1e9 for v in VARIANTS}
    for rnd in range(3):
        for v in VARIANTS:
            ops.set_conv_variant(v)
            time_layer(a, layer, 2)
            best[v] = min(best[v], time_layer(a, layer, 10 if which == "256" else 3))
    for v in VARIANTS:
        res[v].append(f"L{layer}:{fl / best[v] / 1e9:.0f}")
        tot[v][0] += best[v]
        tot[v][1] += fl
for v in VARIANTS:
    print(f"variant {v} {which}: total {tot[v][0]:.3f} ms, {tot[v][1] / tot[v][0] / 1e9:.0f} TF | " + " ".join(res[v]), flush=True)
