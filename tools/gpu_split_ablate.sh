#!/bin/bash
# Timing-only ablations of the f32s stage-1 kernel (conv1_regw_split_kernel, nqa_conv.hip) on the GPU box: the shipped
# library against builds without conv1_1 (NQA_SP_NO_P1), without conv1_2 (NQA_SP_NO_P2), with every store out of range
# (NQA_SP_NO_STORE); results of those are wrong on purpose.  Prints stage 1's time from the layer bench's first line.
# usage: bash tools/gpu_split_ablate.sh
set -e
cd "$(dirname "$0")/.."
run() {  # name, flags...
  name=$1; shift
  if [ -n "$1" ]; then python -m nerf_qa_amd.build --out=libnqa_$name.so "$@" > /dev/null 2>&1; export NQA_LIB=$PWD/nerf_qa_amd/libnqa_$name.so; else unset NQA_LIB; fi
  python - "$name" <<'PY'
import sys, torch
sys.path.insert(0, '.')
from nerf_qa_amd import ops, synth
dev = torch.device("cuda:0")
packed = ops.pack_vgg_weights(synth.vgg16_weights(1234), "f32s").to(dev)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(16, 3, 1080, 1920, device=dev, generator=g)
for _ in range(5):
    ops.conv1_fused(x, packed, "f32s")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv1_fused(x, packed, "f32s")
e1.record()
torch.cuda.synchronize()
print(f"{sys.argv[1]:<16} stage 1 (f32s, 16 images of 1080p): {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)
PY
}
run shipped
run no_p1 -DNQA_SP_NO_P1
run no_p2 -DNQA_SP_NO_P2
run no_store -DNQA_SP_NO_STORE
run no_p1_no_store -DNQA_SP_NO_P1 -DNQA_SP_NO_STORE
