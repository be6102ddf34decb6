"""Per-segment cycle shares of the fused f16 stage-1 kernel from a -DNQA_STAMPS build (NQA_LIB=.../libnqa_stamps.so)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import ops, synth, _lib  # noqa: E402
from nerf_qa_amd._lib import lib, ptr, stream_ptr, check  # noqa: E402
dev = torch.device("cuda:0")
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f16").to(dev)
fn = C.CDLL(_lib.LIB_PATH).nqa_debug_stamps
buf = (C.c_ulonglong * 8)()
for (H, W, N) in ((256, 256, 64), (1080, 1920, 16)):
    x = torch.rand(N, 3, H, W, device=dev)
    out = torch.empty(N, H, W, 64, dtype=torch.float16, device=dev)
    for _ in range(3):
        check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().nqa_conv1_fused(ptr(x), N, H, W, ptr(packed), 2, ptr(out), stream_ptr(dev)))
    e1.record()
    torch.cuda.synchronize()
    fn(buf, 1)
    nt = buf[4]  # wave-tiles
    names = ("barrier", "conv1_2", "conv1_1", "raw")
    tot = sum(buf[i] for i in range(4))
    print(f"{H}x{W} N={N}: {e0.elapsed_time(e1) * 1e3:.0f} us; wave-tiles {nt}; cycles per wave and tile: " +
          " ".join(f"{n}={buf[i] / nt:.0f}" for i, n in enumerate(names)) + f" total={tot / nt:.0f}; conv1_2 of waves 0-3 "
          f"{buf[5] / (nt / 2):.0f}, of waves 4-7 {buf[6] / (nt / 2):.0f}", flush=True)
