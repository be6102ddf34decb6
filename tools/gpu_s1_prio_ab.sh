#!/bin/bash
set -e
cd "$(dirname "$0")/.."
for P in 1 2 3; do python -m nerf_qa_amd.build --out=libnqa_prio$P.so -DNQA_S1_TAIL_PRIO=$P > /dev/null 2>&1; done
for rnd in 1 2; do
for L in "" prio1 prio2 prio3; do
  if [ -n "$L" ]; then export NQA_LIB=$PWD/nerf_qa_amd/libnqa_$L.so; else unset NQA_LIB; fi
  python - "${L:-shipped}" <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from nerf_qa_amd import ops
from nerf_qa_amd.DISTS_pytorch import DISTS
dev = torch.device("cuda:0")
m = DISTS(precision="f16", vgg16_path="synth:1234").to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(8, 3, 1080, 1920, device=dev, generator=g)
y = (x + 0.1 * torch.randn(x.shape, device=dev, generator=g)).clamp_(0, 1)
with torch.no_grad():
    for _ in range(5):
        m(x, y)
    torch.cuda.synchronize()
    ops.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(20):
        m(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
kt = ops.timing_collect()
print(f"{sys.argv[1]:<10} step {dt:7.3f} ms  conv class {kt['conv_igemm'][1] / 20:7.3f}", flush=True)
PY
done
done
