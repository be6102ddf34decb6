"""The HBM yardstick of the pool+statistics pass (development aid, GPU box): hand-written 16 B/lane streaming
kernels (tools/hip/stream_bw.hip, compiled here with hipcc) -- copy, read-4-write-1 (the pool pass's mix) and
read-only -- at the pass's own size, next to the pass itself on the same box in the same process."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_qa_amd import ops  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

so = "/tmp/libstream_bw.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                os.path.join(ROOT, "tools/hip/stream_bw.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.stream_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
lib.stream_run2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


nbytes = 2 << 30  # 2 GiB read (the largest tap of a B=8 1080p step in f16 is 4.2 GB)
src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
names = {0: "copy (1 read : 1 write)", 1: "read 4 : write 1", 2: "read only", 3: "fill (write only)"}
moved = {0: 2 * nbytes, 1: nbytes + nbytes // 4, 2: nbytes, 3: nbytes}
cus = torch.cuda.get_device_properties(dev).multi_processor_count
for kind in (0, 1, 2, 3):
    for nt in (0, 1):
        if kind == 3 and nt:
            continue
        for unroll in ((1, 2, 4) if kind == 1 else (4,)):
            for blocks in (cus * 8, cus * 32, 32768):
                ms = timed(lambda: lib.stream_run2(kind, nt, unroll, src.data_ptr(), dst.data_ptr(), nbytes, blocks, st))
                print(f"{names[kind]:<24} nt={nt} rows/lane={4 * unroll if kind == 1 else 4:2d} blocks={blocks:6d}: {ms:.3f} ms  "
                      f"{moved[kind] / ms / 1e6:7.1f} GB/s", flush=True)
# producer -> consumer through the 256 MiB Infinity Cache: a buffer is written (fill) and then read by the next kernel;
# the read alone is timed.  If a just-written slab is served on-die, the read runs well above the HBM rate for slabs
# that fit and falls back to it for slabs that do not.
for mib in (32, 64, 128, 192, 256, 384, 512, 1024):
    nb = mib << 20
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for k in range(12):
        lib.stream_run2(3, k, 4, src.data_ptr(), dst.data_ptr(), nb, cus * 8, st)
        e[k][0].record()
        lib.stream_run2(2, 0, 4, dst.data_ptr(), src.data_ptr(), nb, cus * 8, st)
        e[k][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in e[2:])
    ms = ts[len(ts) // 2]
    print(f"read right after fill, {mib:5d} MiB: {ms:.4f} ms  {nb / ms / 1e6:8.1f} GB/s", flush=True)
t = timed(lambda: dst.copy_(src))
print(f"torch copy_: {t:.3f} ms {2 * nbytes / t / 1e6:.1f} GB/s")
del src, dst

# the pass itself: DISTS B=8 1080p, f16 and f32s, pool+statistics time per step from the library's event ring
for prec in ("f16", "f32s"):
    m = DISTS(precision=prec, vgg16_path="synth:1234").to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(8, 3, 1080, 1920, device=dev, generator=g)
    y = (x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1)
    with torch.no_grad():
        for _ in range(2):
            m(x, y)
        torch.cuda.synchronize()
        ops.timing_enable(True)
        for _ in range(5):
            m(x, y)
        torch.cuda.synchronize()
    kt = ops.timing_collect()
    ops.timing_enable(False)
    n, ms = kt["l2pool"]
    esz = 2 if prec == "f16" else 4
    alg = sum(16 * (h * w * c * esz + ((h + 1) // 2) * ((w + 1) // 2) * c * esz)
              for (h, w), c in zip(ops.pyramid_dims(1080, 1920)[:4], ops.CHNS[1:5]))
    print(f"pool_stats {prec}: {ms / 5:.3f} ms per step ({n // 5} launches), algorithmic {alg / 1e9:.2f} GB -> "
          f"{alg / (ms / 5) / 1e6:.1f} GB/s")
    del m
    torch.cuda.empty_cache()
