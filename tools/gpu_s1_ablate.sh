#!/bin/bash
# Timing-only ablations of the fused stage-1 kernel (nqa_conv1_pool.hip) on the GPU box: the shipped library against
# builds whose tail waves skip the tile work (statistics, pool, stores), conv1_1, or both, and builds whose conv waves
# skip their MFMAs (results of those are wrong on
# purpose); prints the conv class time of a DISTS B=8 1080p step for each.  usage: bash tools/gpu_s1_ablate.sh
set -e
cd "$(dirname "$0")/.."
run() {  # name, flags...
  name=$1; shift
  if [ -n "$1" ]; then python -m nerf_qa_amd.build --out=libnqa_$name.so "$@" > /dev/null 2>&1; export NQA_LIB=$PWD/nerf_qa_amd/libnqa_$name.so; else unset NQA_LIB; fi
  python - "$name" <<'PY'
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
    for _ in range(3):
        m(x, y)
    torch.cuda.synchronize()
    ops.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(10):
        m(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
kt = ops.timing_collect()
print(f"{sys.argv[1]:<28} step {dt:7.3f} ms  conv class {kt['conv_igemm'][1] / 10:7.3f}  pool class {kt['l2pool'][1] / 10:6.3f}", flush=True)
PY
}
run shipped
run no_epi -DNQA_S1_NO_EPI
run no_c11 -DNQA_S1_NO_C11
run no_epi_no_c11 -DNQA_S1_NO_EPI -DNQA_S1_NO_C11
run no_mfma -DNQA_S1_NO_MFMA
run no_mfma_no_c11 -DNQA_S1_NO_MFMA -DNQA_S1_NO_C11
