"""Read the stage-loop cycle stamps of a -DNQA_STAMPS build (NQA_LIB=.../libnqa_stamps.so)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth, _lib  # noqa: E402
import os
dev = torch.device("cuda:0")
PREC = os.environ.get("NQA_TOOL_PREC", "f16")
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), PREC).to(dev)
L = _lib.lib()
fn = C.CDLL(_lib.LIB_PATH).nqa_debug_stamps
buf = (C.c_ulonglong * 8)()
which = sys.argv[1] if len(sys.argv) > 1 else "256"
HH, WW, NN = (256, 256, 64) if which == "256" else (1080, 1920, 16)
dims = ops.pyramid_dims(HH, WW)
for layer, variant in ((1, 0), (2, 0), (3, 0), (4, 1), (4, 0), (5, 1), (8, 1), (8, 0), (10, 0)):
    ops.set_conv_variant(variant)
    h, w = dims[ops.CONV_STAGE[layer]]
    a = (torch.rand(NN, h, w, ops.CONV_CIN[layer], device=dev) - 0.5).clamp_min(0).to(_lib.PREC_DTYPE[_lib.prec_id(PREC)])
    for _ in range(3):
        ops.conv3x3_relu(a, layer, packed, PREC)
    torch.cuda.synchronize()
    fn(buf, 1)
    ops.conv3x3_relu(a, layer, packed, PREC)
    torch.cuda.synchronize()
    fn(buf, 1)
    nst = buf[4]
    names = ("dma_wait", "barrier", "dma_issue", "compute")
    tot = sum(buf[i] for i in range(4))
    print(f"{which} layer {layer} variant {variant}: wave-stages={nst} per-stage cycles: " +
          " ".join(f"{n}={buf[i] / nst:.0f}" for i, n in enumerate(names)) + f" total={tot / nst:.0f}")
