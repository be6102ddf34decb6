"""One-rank RCCL sanity check of exactly the calls bench.py / sharding.py make at N > 1 (init with device_id,
barrier, all_reduce MAX, all_gather_into_tensor through sharding.gather_scores)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import sharding  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
scores = torch.arange(10, dtype=torch.float32, device=dev)
out = sharding.gather_scores(scores, 10)
assert torch.equal(out, scores) and t.item() == 1.5
dist.destroy_process_group()
print("RCCL one-rank check ok")
