"""What this box's HBM gives plain streaming kernels (torch device copy = 1 read + 1 write stream; sum = read only):
the yardstick for the pool+statistics pass (development aid)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 31  # 4 GiB of half -> 2^31 elements
a = torch.empty(n, dtype=torch.float16, device=dev).normal_()
b = torch.empty_like(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, bytes_ in (("copy (read + write)", lambda: b.copy_(a), 2 * a.numel() * 2),
                         ("sum (read only)", lambda: a.sum(dtype=torch.float32), a.numel() * 2)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms, {bytes_ / ms / 1e9:.2f} TB/s", flush=True)
c = torch.empty(1 << 31, dtype=torch.float32, device=dev)  # 8 GiB: the size of conv1_1's f32s output at 1080p B=8
for _ in range(2):
    c.zero_()
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    c.zero_()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fill (write only): {ms:.3f} ms, {c.numel() * 4 / ms / 1e9:.2f} TB/s", flush=True)
