"""BASELINE configs[3]: a 10 000-frame synthetic 256x256 video scored through the harness
(video.score_video: uint8 frames -> ToTensor on the device -> batched DISTS -> per-video columns), frames
sharded over the ranks with ONE all-gather of the scores.  Strong scaling: the video is fixed.

    python tools/video10k.py                                            # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/video10k.py
"""
import json
import os
import sys
import time
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import video  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

N_FRAMES = int(os.environ.get("NQA_VIDEO_FRAMES", "10000"))
world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
dev_index = local % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("NQA_DIST_BACKEND", "nccl")
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
g = torch.Generator(device=dev).manual_seed(7)  # the same video on every rank
ref = torch.randint(0, 256, (N_FRAMES, 256, 256, 3), dtype=torch.uint8, device=dev, generator=g)
noise = torch.randint(-12, 13, ref.shape, dtype=torch.int16, device=dev, generator=g)
ren = (ref.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
del noise
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = DISTS().to(dev).eval()
video.score_video(ref[:64], ren[:64], dists_model=m, batch_size=32, policy="full")  # warm-up
torch.cuda.synchronize(dev)
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
cols = video.score_video(ref, ren, dists_model=m, batch_size=32, policy="full")
torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
if rank == 0:
    print(json.dumps({"metric": "DISTS frames/s, 10k-frame 256x256 video (BASELINE configs[3])",
                      "value": round(N_FRAMES / dt, 1), "unit": "frame-pairs/s", "n_gpus": world, "scaling": "strong",
                      "seconds": round(dt, 4), "columns": {k: round(v, 6) for k, v in cols.items()}}))
if world > 1:
    dist.destroy_process_group()
