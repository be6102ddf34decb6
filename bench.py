#!/usr/bin/env python3
"""DISTS frame-pairs/s on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 256|1080p] [--precision f16]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one DISTS.forward over one batch of synthetic frame pairs that are already
resident in HBM (default workload = BASELINE.json configs[1]: 32 pairs of 256x256 per GPU).
Frames shard across ranks with no data-path collective; the only exchange is one
all-gather of the per-frame scores after the last step (inside the timed region).  Rank 0
prints ONE JSON line.  The `roofline` object is for the dominant kernel (the MFMA
implicit-GEMM conv3x3): algorithmic FLOPs of layers 1..12 per step / the HIP-event time of
those launches, measured inside the timed region.  `cpu_baseline` times the CPU oracle
(the reference's arithmetic, bit-identical to it in the authoring container) on a bounded
sample on this box's host cores.
"""
import argparse
import json
import os
import statistics
import sys
import time
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerf_qa_amd import ops, sharding, synth  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

WORKLOADS = {
    "256": dict(name="configs[1]: B=32 256x256 synthetic frame pairs per GPU", B=32, H=256, W=256, metric="DISTS"),
    "1080p": dict(name="configs[2]: B=8 1920x1080 synthetic frame pairs per GPU", B=8, H=1080, W=1920, metric="DISTS"),
    "adists1080p": dict(name="configs[4]: A-DISTS, B=8 1920x1080 synthetic frame pairs per GPU", B=8, H=1080, W=1920,
                        metric="A-DISTS"),
    "adists256": dict(name="A-DISTS, B=32 256x256 synthetic frame pairs per GPU", B=32, H=256, W=256,
                      metric="A-DISTS"),
}
PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3,  # dense MFMA, MI355X_MICROARCH.md
               "f32s": 2500.0 / 3}  # split-f16: three half MFMAs per algorithmic product


def conv_flops_per_image(h, w):
    """(igemm layers 1..12, conv1_1) algorithmic FLOPs = 2*9*Cin*Cout*Hk*Wk summed."""
    dims = ops.pyramid_dims(h, w)
    ig = 0
    for li in range(1, 13):
        hk, wk = dims[ops.CONV_STAGE[li]]
        ig += 2 * 9 * ops.CONV_CIN[li] * ops.CONV_COUT[li] * hk * wk
    return ig, 2 * 9 * 3 * 64 * h * w


def pool_bytes_per_image(h, w, esz):
    """Algorithmic HBM bytes of the pool+statistics pass per image: taps 1..4 read once, a quarter written
    (SURVEY 8d, L2-pool row); esz = bytes per stored activation."""
    dims = ops.pyramid_dims(h, w)
    total = 0
    for k in range(4):
        hk, wk = dims[k]
        c = ops.CHNS[k + 1]
        total += hk * wk * c * esz + ((hk + 1) // 2) * ((wk + 1) // 2) * c * esz
    return total


def hbm_roofline(launches_ms, h, w, b, steps, prec):
    """Secondary roofline object: the HBM-bound pool+statistics kernel (algorithmic bytes / HIP-event time)."""
    n, ms = launches_ms
    esz = 2 if prec in ("f16", "bf16") else 4
    ach = pool_bytes_per_image(h, w, esz) * 2 * b * steps / (ms * 1e-3) / 1e9 if ms > 0 else None
    return {"kernel": "pool_stats_kernel (L2-pool + the five statistics sums of taps 1..4, one pass)", "bound": "hbm",
            "achieved": round(ach, 1) if ach else None, "peak": 8000.0, "unit": "GB/s",
            "frac": round(ach / 8000.0, 4) if ach else None, "traffic": None, "launches": n}


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota if there is one (the GPU box
    shows 256 logical CPUs but grants a 16-core share), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(h, w, budget_s=20.0, adists=False):
    """The oracle (kind "port") on the host cores: frame-pairs/s on a bounded sample."""
    from oracle import adists_oracle, dists_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234))
    ab = DISTS_alpha_beta()
    if adists:
        run = lambda a, b: adists_oracle.adists(a, b, convs)  # noqa: E731
    else:
        run = lambda a, b: dists_oracle.dists(a, b, convs, *ab)  # noqa: E731
    # size the sample from one warm-up pair so the whole leg stays near the budget
    xn, yn = synth.frame_batch([0], h, w)
    x1, y1 = torch.from_numpy(xn), torch.from_numpy(yn)
    t0 = time.perf_counter()
    run(x1, y1)
    t_one = time.perf_counter() - t0
    n = max(1, min(8, int(budget_s / 3.0 / max(t_one, 1e-3))))
    x, y = x1.repeat(n, 1, 1, 1), y1.repeat(n, 1, 1, 1)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(x, y)
        times.append(time.perf_counter() - t0)
        if sum(times) > budget_s:
            break
    return {"value": round(n / statistics.median(times), 4), "unit": "frame-pairs/s", "cores": cores,
            "kind": "port", "sample": f"{n} pair(s) of {h}x{w}, CPU oracle fp32, median of {len(times)} after 1 warm-up"}


def DISTS_alpha_beta():
    import numpy as np
    d = np.load(os.path.join(ROOT, "nerf_qa_amd", "data", "dists_alpha_beta.npz"))
    return torch.from_numpy(d["alpha"]).view(1, -1, 1, 1), torch.from_numpy(d["beta"]).view(1, -1, 1, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="256",
                    help="256 (default, BASELINE configs[1]) | 1080p | adists1080p | adists256")
    ap.add_argument("--precision", default=None, help="f16 (DISTS default), f32s (A-DISTS default), f32, bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override pairs per GPU per step (experiments only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # one rank per GPU; the modulo only matters when the N>1 path is rehearsed on a 1-GPU box
    # (NQA_DIST_BACKEND=gloo, several ranks sharing cuda:0)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NQA_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    wl = WORKLOADS[args.workload]
    B, H, W = args.batch or wl["B"], wl["H"], wl["W"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if wl["metric"] == "A-DISTS":
            from nerf_qa_amd.ADISTS import ADISTS
            net = ADISTS(precision=args.precision).to(dev).eval()
            model = lambda a, b: net(a, b, as_loss=False)  # noqa: E731  (x = reference frame drives ps / weights)
            model.precision, model.vgg_source = net.precision, net.vgg_source
        else:
            model = DISTS(precision=args.precision).to(dev).eval()
    prec = model.precision_for(H, W) if hasattr(model, "precision_for") else model.precision

    # synthetic frames generated on the device (no host I/O in the timed region); each rank
    # seeds with its rank so shards differ
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.rand(B, 3, H, W, device=dev, generator=g)
    y = (x + 0.1 * torch.randn(B, 3, H, W, device=dev, generator=g)).clamp_(0, 1)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    scores = torch.empty(args.steps * B, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for _ in range(args.warmup):
            model(x, y)
        if world > 1:  # warm the collective too
            sharding.gather_scores(scores, world * scores.numel())
        sync()
        ops.timing_enable(True)
        t0 = time.perf_counter()
        for k in range(args.steps):
            scores[k * B:(k + 1) * B] = model(x, y)
        all_scores = sharding.gather_scores(scores, world * scores.numel()) if world > 1 else scores
        sync()
        dt = time.perf_counter() - t0
    ktimes = ops.timing_collect()
    ops.timing_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(all_scores).all()

    if rank == 0:
        ig_flops, c1_flops = conv_flops_per_image(H, W)
        n_ig, ms_ig = ktimes["conv_igemm"]
        launches_per_step = 12
        steps_timed = n_ig / launches_per_step if n_ig else 0
        achieved = (ig_flops * 2 * B * steps_timed) / (ms_ig * 1e-3) / 1e12 if ms_ig > 0 else None
        peak = PEAK_TFLOPS[prec]
        out = {
            "metric": wl["metric"] + " frame-pairs/s",
            "value": round(world * B * args.steps / dt, 2),
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": prec,
            "data": "synthetic",
            "config": {"workload": wl["name"], "pairs_per_gpu_per_step": B, "height": H, "width": W,
                       "vgg_weights": model.vgg_source, "sharding": f"frames/{world} ranks, one all-gather of scores"},
            "roofline": {
                "kernel": "conv3x3_igemm_kernel (VGG layers 1..12, MFMA implicit GEMM)",
                "bound": "mfma", "achieved": round(achieved, 2) if achieved else None, "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4) if achieved else None, "traffic": None,
                "launches": n_ig, "avg_launch_ms": round(ms_ig / n_ig, 5) if n_ig else None,
                "flop_per_launch_avg": round(ig_flops * 2 * B / launches_per_step),
            },
            "kernel_ms_per_step": {k: round(v[1] / max(steps_timed, 1), 4) for k, v in ktimes.items() if v[0]},
            "roofline_hbm": hbm_roofline(ktimes["l2pool"], H, W, B, steps_timed, prec),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, W, adists=wl["metric"] == "A-DISTS")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
