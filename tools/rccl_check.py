"""One-rank RCCL sanity check of exactly the calls bench.py / sharding.py make at N > 1 (init with device_id,
barrier, all_reduce MAX, all_gather_into_tensor through sharding.gather_scores, and the object broadcast of DISTS'
calibration verdict through sharding.agree_precision on a real `auto` module)."""
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
box = [("f16", {"choice": "f16", "pairs": 256})]
dist.broadcast_object_list(box, src=0, device=dev)
assert box[0][0] == "f16" and box[0][1]["pairs"] == 256
os.environ.setdefault("NQA_CAL_CACHE", "off")
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
net = DISTS(vgg16_path="synth:1234").to(dev).eval()
mode = sharding.agree_precision(net, 160, 192, dev)  # class 0: 384 small pairs, ~2 s
assert mode == net.precision_for(160, 192, dev) and net._agreed_report["agreed_over_ranks"] == 1, (mode, net._agreed_report)
dist.destroy_process_group()
print("RCCL one-rank check ok")
