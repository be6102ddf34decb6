#!/usr/bin/env python3
"""DISTS / A-DISTS frame-pairs/s on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p|256|adists1080p|adists256|video10k]
                  [--precision f16|f32s|f32|bf16] [--only]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts that torch.distributed.run
command itself as a CHILD process (before anything touches the GPU), relays rank 0's JSON line and exits with the
child's code; under an external launcher (WORLD_SIZE set) it is one of the N ranks.

The line's `value` is the headline workload, BASELINE.json configs[2]: B=8 pairs of 1920x1080 per GPU per
step through DISTS.forward in its shipped precision mode -- "auto": the fastest of f16 / f16w / f32m4 / f32m / f32m2 /
f32s that the module's one-time calibration of its VGG weights admits (384 synthetic pairs through every rung; a mode is
admitted when its deviation from f32s stays well inside the 1e-4 bar AND looks like noise rather than outliers,
DISTS_pt.py header), per frame-size class.  For the stand-in weights used here (`synth:1234`, gain 1.0) and 1080p frames
that is plain f16, with stage 1 and conv2_2 pooling and summing their own taps (taps 1-2 never reach HBM at full
resolution; every frame below 0.9 Mpx runs f32s by default since the calibration has NeRF-like content); weights with
ImageNet-like activation growth (gain 1.3) calibrate to f32m at 1080p and the stress set (gain 1.6) to f32s -- the line
says what ran in `dtype`,
`config.precision`, `config.auto_calibration`, and carries the other two defaults as companions.  Frames are resident in HBM.  Frames
shard across ranks with no data-path collective; the only exchange is ONE all-gather of the per-frame scores
after the last step (inside the timed region) -- weak scaling, K steps of 8 pairs on every GPU.

At N=1 the same JSON line also carries, under "workloads", the rest of the metric ("1080p & 256^2", DISTS
and A-DISTS) measured the same way in the same process: 1080p in f16 (the opt-in fast mode: one MFMA per product; NOT
what `auto` admits for every weight set), in f32m / f32m2 and in f32s (float32 activations, split-f16 products: the
reference's own precision class, what `auto` runs when nothing faster is admitted), the shipped default ON THE OTHER TWO
PINNED WEIGHT SETS (`1080p/auto@gain1.3`: the ImageNet-like activation magnitude, `1080p/auto@gain1.6`: the stress set --
whatever rung each calibrates to, with its calibration report), exact `f32` (the only mode with the reference's 24-bit
products; peak 157.3 TF), 256x256 B=32 (configs[1]) in f16 / f32m / f32s, and A-DISTS at 1080p B=8 (configs[4], f32s).
`config.default_by_weight_set` puts the three defaults side by side.  Every entry has its own
`roofline` (the MFMA conv stack: algorithmic FLOPs of layers 1..12 / the HIP-event time of those launches, measured
inside the timed region on the launch stream; `peak` is always the guide's dense 2.5 PFLOP/s f16 figure, and for
f32m / f32s both the algorithmic and the issued-MFMA fraction are given) and `roofline_hbm` (the HBM-bound L2-pool +
statistics pass over the taps the conv kernels do not close themselves -- `fused_taps` names the others -- against
the guide's 8 TB/s and against the best hand-written stream of the same read / write mix).  `roofline.traffic` comes
from the committed rocprofv3 PMC summary of this same command (profiles/r04_traffic.json), per launch, and is withheld
when the library that ran is not the profiled build.

`--workload video10k` is BASELINE.json configs[3]: a 10 000-frame 1080p video whose frames are generated on
the device per batch from seed = frame index, sharded over the ranks (strong scaling), one all-gather.

`cpu_baseline` times the CPU oracle (the reference's arithmetic, bit-identical to it in the authoring
container) on a bounded sample of the headline workload on this box's host cores.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nerf_qa_amd import ops, sharding, synth, video  # noqa: E402
from nerf_qa_amd.ADISTS import ADISTS  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402

VGG = "synth:1234"  # no ImageNet checkpoint offline: the deterministic stand-in weights, asked for explicitly
WORKLOADS = {
    "1080p": dict(name="configs[2]: B=8 1920x1080 synthetic frame pairs per GPU", B=8, H=1080, W=1920, metric="DISTS"),
    "256": dict(name="configs[1]: B=32 256x256 synthetic frame pairs per GPU", B=32, H=256, W=256, metric="DISTS"),
    "adists1080p": dict(name="configs[4]: A-DISTS, B=8 1920x1080 synthetic frame pairs per GPU", B=8, H=1080, W=1920,
                        metric="A-DISTS"),
    "adists256": dict(name="A-DISTS, B=32 256x256 synthetic frame pairs per GPU", B=32, H=256, W=256,
                      metric="A-DISTS"),
    "video10k": dict(name="configs[3]: 10k-frame 1920x1080 synthetic video, frames generated on the device from "
                          "seed = frame index, sharded over the ranks", B=8, H=1080, W=1920, metric="DISTS"),
}
# what the N=1 line measures beside the headline (workload key, precision)
# what the N=1 line measures beside the headline: (workload key, precision[, VGG weight spec]); precision None = the
# shipped default (`auto`) calibrated on that weight set
COMPANIONS = (("1080p", None, "synth:1234:1.3"), ("1080p", None, "synth:1234:1.6"),
              ("1080p", "f16"), ("1080p", "f16w"), ("1080p", "f32m"), ("1080p", "f32m2"), ("1080p", "f32s"), ("1080p", "f32"),
              ("256", "f16"), ("256", "f16w"), ("256", "f32m"), ("256", "f32s"),
              ("adists1080p", "f32s"))  # (the one the headline itself ran in is skipped)
PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3   # exact-f32 MFMA
PEAK_HBM_GBS = 8000.0
MEASURED_STREAM_GBS = 6200.0  # best 4-read:1-write stream (non-temporal loads), profiles/r04_stream_bw.txt
MFMA_PER_PRODUCT = {"f16": 1, "bf16": 1, "f32s": 3, "f32": 1}  # f32m: 2 for conv layers 1..6, 3 for 7..12
MIXED_LAST_2TERM_LAYER = {"f32m": 6, "f32m2": 3, "f32m4": 9, "f16w": 12}
MIXED_STAGES = {"f32m": 3, "f32m2": 2, "f32m4": 4, "f16w": 5}
TRAFFIC_FILE = "profiles/r04_traffic.json"
ONE_STREAM_NOTE = ("; A-DISTS: `value` runs each batch as two half-batches on two HIP streams (the shipped form), the kernel "
                   "times behind this entry come from a second pass of the same steps on ONE stream, where they are additive")
AUTO_REPORT = {}  # DISTS' one-time precision calibration (what `auto`, the shipped default, chose and on what evidence)


def conv_flops_per_image(h, w, prec=None):
    """(igemm layers 1..12, conv1_1) algorithmic FLOPs = 2*9*Cin*Cout*Hk*Wk summed; with `prec` the
    first figure is the ISSUED f16-MFMA FLOPs instead (terms per product: f32s 3; f32m 2 for layers 1..6, 3 behind)."""
    dims = ops.pyramid_dims(h, w)
    ig = 0
    for li in range(1, 13):
        hk, wk = dims[ops.CONV_STAGE[li]]
        terms = 1 if prec is None else (
            (2 if li <= MIXED_LAST_2TERM_LAYER[prec] else 3) if prec in MIXED_LAST_2TERM_LAYER else MFMA_PER_PRODUCT.get(prec, 1))
        ig += terms * 2 * 9 * ops.CONV_CIN[li] * ops.CONV_COUT[li] * hk * wk
    return ig, 2 * 9 * 3 * 64 * h * w


def tap_elem_bytes(prec, k):
    """(bytes per element of tapped map k+1, bytes per element of its pooled map) in mode `prec`."""
    if prec in MIXED_STAGES:  # half taps up to the last two-term stage (whose pool writes split16 records), float behind
        ms = MIXED_STAGES[prec]
        return (2, 2 if k < ms - 1 else 4) if k < ms else (4, 4)
    e = 2 if prec in ("f16", "bf16") else 4
    return e, e


def pool_bytes_per_image(h, w, prec, taps=(1, 2, 3, 4)):
    """Algorithmic HBM bytes of the pool+statistics pass per image: each tap in `taps` read once, a quarter written
    (SURVEY 8d, L2-pool row), in the element sizes the mode stores them in.  Taps the forward closes inside their
    conv kernel (ops.dists_fused_taps) never go through this pass and are left out by the caller."""
    dims = ops.pyramid_dims(h, w)
    total = 0
    for k in (t - 1 for t in taps):
        hk, wk = dims[k]
        c = ops.CHNS[k + 1]
        ein, eout = tap_elem_bytes(prec, k)
        total += hk * wk * c * ein + ((hk + 1) // 2) * ((wk + 1) // 2) * c * eout
    return total


def load_traffic():
    """Committed PMC summary (rocprofv3 --pmc passes of this command, corrected as MI355X_MICROARCH.md's HBM
    section prescribes: FETCH_SIZE doubled on gfx950): {workload/prec: {"conv": bytes per launch, "pool": ...}}.
    The figures belong to the library SOURCES named in the file ("lib_sha16" = nerf_qa_amd.build.source_hash());
    when the tree's sources differ the bytes are withheld and the line says why, so stale bytes never pass as fresh."""
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
    except (OSError, ValueError):
        return {}
    try:
        from nerf_qa_amd import build as nqa_build
        t["_lib_matches"] = nqa_build.source_hash() == t.get("lib_sha16") and not os.environ.get("NQA_LIB")
    except OSError:
        t["_lib_matches"] = False
    return t


def traffic_for(traffic, key):
    """The committed PMC bytes per launch of workload `key`, tagged with where they come from and whether the
    profiled library is the one that just ran (None + the reason when there is nothing to show)."""
    t = dict(traffic.get(key, {}))
    if not t:
        t["_source"] = f"none: {TRAFFIC_FILE} has no entry for {key}"
    elif not traffic.get("_lib_matches"):
        t = {"_source": f"withheld: {TRAFFIC_FILE} was measured on library build {traffic.get('lib_sha16')}, not the one "
                        "that ran this line"}
    else:
        t["_source"] = f"{TRAFFIC_FILE} (rocprofv3 --pmc passes of this command on library build {traffic.get('lib_sha16')})"
    return t


def rooflines(ktimes, h, w, b, prec, traffic, fused=None, steps=None):
    """roofline (MFMA conv stack) and roofline_hbm (pool+statistics) from the HIP-event times of one timed run of
    `steps` steps (None: inferred from the launch count at 12 conv launches per step -- one pyramid pass over the batch;
    A-DISTS runs a batch as two half-batches on two streams, i.e. 24 launches per step, so its callers say how many)."""
    ig_flops, _ = conv_flops_per_image(h, w)
    issued_flops, _ = conv_flops_per_image(h, w, prec)
    n_ig, ms_ig = ktimes["conv_igemm"]
    if steps is None:
        steps = n_ig / 12 if n_ig else 0
    per_step = n_ig / steps if steps and n_ig else 12  # launches of class conv_igemm per step
    ach = ig_flops * 2 * b * steps / (ms_ig * 1e-3) / 1e12 if ms_ig > 0 else None
    peak = PEAK_F32_TFLOPS if prec == "f32" else PEAK_F16_TFLOPS
    roof = {
        "kernel": "VGG conv layers 1..12 on MFMA: conv1_pool_kernel (stage 1 + pool + statistics, f16) or conv1_regw_kernel, "
                  "conv3x3_regw_kernel (conv2_1), conv3x3_regw128_pool_kernel (conv2_2 + pool + statistics, f16 stage) or "
                  "conv3x3_regw128_kernel (conv2_2, conv3_1), conv3x3_igemm_kernel (the rest; all layers in f32; f32s: "
                  "conv1_regw_split_kernel for stage 1, igemm behind; f32m: the two-term instances of the first three for "
                  "layers 1..4, two-term igemm for 5..6, f32s igemm for 7..12)",
        "bound": "mfma", "achieved": round(ach, 2) if ach else None, "peak": peak, "unit": "TFLOP/s",
        "frac": round(ach / peak, 4) if ach else None,
        "traffic": traffic.get("conv"), "traffic_source": traffic.get("_source"),
        "launches": n_ig, "avg_launch_ms": round(ms_ig / n_ig, 5) if n_ig else None,
        "flop_per_launch_avg": round(ig_flops * 2 * b / per_step),
        "note": "achieved = algorithmic FLOPs / HIP-event time; peak = dense f16 MFMA (2.5 PF)" if prec != "f32" else
                "exact-f32 MFMA; peak = 157.3 TF",
    }
    if (prec == "f32s" or prec in MIXED_STAGES) and ach:
        issued = ach * issued_flops / ig_flops
        roof["mfma_issued_tflops"] = round(issued, 1)
        roof["frac_of_issued_mfma"] = round(issued / peak, 4)
        roof["note"] = ("f32s issues three f16 MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi), f32m two for conv "
                        "layers 1..6 (activation x weight-hi, x weight-lo) and three behind: `frac` is algorithmic "
                        "FLOP/s over the 2.5 PF f16 peak, `frac_of_issued_mfma` is the matrix cores' load")
    n_p, ms_p = ktimes["l2pool"]
    fused = ops.dists_fused_taps(b, h, w, prec) if fused is None else fused
    left = tuple(t for t in (1, 2, 3, 4) if t not in fused)
    pool_b = pool_bytes_per_image(h, w, prec, left)
    ach_b = pool_b * 2 * b * steps / (ms_p * 1e-3) / 1e9 if ms_p > 0 and steps and left else None
    hbm = {"kernel": f"pool_stats_kernel (L2-pool + the five statistics sums of taps {list(left)}, one pass each; taps "
                     f"{list(fused)} are pooled and summed inside their conv kernel and never reach HBM at full resolution)",
           "bound": "hbm", "achieved": round(ach_b, 1) if ach_b else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": round(ach_b / PEAK_HBM_GBS, 4) if ach_b else None,
           "frac_of_measured_stream": round(ach_b / MEASURED_STREAM_GBS, 4) if ach_b else None,
           "measured_stream_note": f"{MEASURED_STREAM_GBS:.0f} GB/s = the best 4-read:1-write streaming kernel on this "
                                   "hardware (profiles/r04_stream_bw.txt), the attainable ceiling of this access pattern",
           "traffic": traffic.get("pool"), "traffic_source": traffic.get("_source"), "launches": n_p,
           "bytes_per_launch_avg": round(pool_b * 2 * b / max(len(left), 1)),
           "fused_taps": list(fused), "seam_launches": ktimes.get("pool_seam", (0, 0.0))[0]}
    kms = {k: round(v[1] / max(steps, 1), 4) for k, v in ktimes.items() if v[0]}
    return roof, hbm, kms


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


def DISTS_alpha_beta():
    import numpy as np
    d = np.load(os.path.join(ROOT, "nerf_qa_amd", "data", "dists_alpha_beta.npz"))
    return torch.from_numpy(d["alpha"]).view(1, -1, 1, 1), torch.from_numpy(d["beta"]).view(1, -1, 1, 1)


def cpu_baseline(h, w, budget_s=24.0, adists=False):
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
        if sum(times) + t_one > budget_s:
            break
    return {"value": round(n / statistics.median(times), 4), "unit": "frame-pairs/s", "cores": cores,
            "kind": "port", "sample": f"{n} pair(s) of {h}x{w}, CPU oracle fp32 ({'A-DISTS' if adists else 'DISTS'}), "
                                      f"median of {len(times)} after 1 warm-up"}


def make_model(metric, precision, dev, h, w, vgg=VGG):
    """-> (callable(ref, render) -> (B,) scores, the precision mode it runs this frame size in, weight source)."""
    if metric == "A-DISTS":
        net = ADISTS(precision=precision, vgg16_path=vgg).to(dev).eval()
        return (lambda a, b: net(a, b, as_loss=False)), net.precision_for(h, w), net.vgg_source  # x = reference frame
    net = DISTS(precision=precision, vgg16_path=vgg).to(dev).eval()
    # "auto" (the default) takes its verdict from the calibration file when these weights were calibrated on this device
    # model and library build before, else measures every rung against f32s once; under a process group rank 0 does
    # that and broadcasts the report (one mode per video whatever N, one calibration per job)
    prec = sharding.agree_precision(net, h, w, dev)
    if net.precision == "auto":
        rep = getattr(net, "_agreed_report", None) or net.calibrate(dev, h, w)
        AUTO_REPORT.clear()
        AUTO_REPORT.update({k: (round(v, 9) if isinstance(v, float) else v) for k, v in rep.items()})
    return net, prec, net.vgg_source


def synth_frames(b, h, w, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.rand(b, 3, h, w, device=dev, generator=g)
    y = (x + 0.1 * torch.randn(b, 3, h, w, device=dev, generator=g)).clamp_(0, 1)
    return x, y


def run_workload(key, precision, steps, warmup, dev, world, rank, batch=0, vgg=VGG):
    """K timed steps of one workload on this rank; returns (dt max-over-ranks, scores, ktimes, prec, B, H, W, src)."""
    wl = WORKLOADS[key]
    B, H, W = batch or wl["B"], wl["H"], wl["W"]
    model, prec, src = make_model(wl["metric"], precision, dev, H, W, vgg)
    # synthetic frames generated on the device (no host I/O anywhere); each rank seeds with its rank so shards differ
    x, y = synth_frames(B, H, W, dev, 1000 + rank)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    scores = torch.empty(steps * B, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for _ in range(warmup):
            model(x, y)
        if world > 1:  # warm the collective too
            sharding.gather_scores(scores, world * scores.numel())
        sync()
        ops.timing_enable(not os.environ.get("NQA_BENCH_NO_EVENTS"))  # (development: what the event pairs cost)
        t0 = time.perf_counter()
        for k in range(steps):
            scores[k * B:(k + 1) * B] = model(x, y)
        all_scores = sharding.gather_scores(scores, world * scores.numel()) if world > 1 else scores
        sync()
        dt = time.perf_counter() - t0
    ktimes = ops.timing_collect()
    ops.timing_enable(False)
    assert torch.isfinite(all_scores).all()
    if wl["metric"] == "A-DISTS" and os.environ.get("NQA_ADISTS_STREAMS", "2") != "1":
        # A-DISTS runs a batch as two half-batches on two HIP streams: launches that share the chip have HIP-event
        # durations that are not additive (each runs slower for the company, or they do not overlap at all -- both
        # were seen).  `value` is the shipped two-stream form, timed above; the per-kernel times behind `roofline` come
        # from a second pass of the same K steps on ONE stream (12 conv launches per step, additive).
        os.environ["NQA_ADISTS_STREAMS"] = "1"
        try:
            with torch.no_grad():
                model(x, y)
                torch.cuda.synchronize(dev)
                ops.timing_enable(True)
                for k in range(steps):
                    model(x, y)
                torch.cuda.synchronize(dev)
            ktimes = ops.timing_collect()
            ops.timing_enable(False)
        finally:
            del os.environ["NQA_ADISTS_STREAMS"]
    del x, y
    return dt, ktimes, prec, B, H, W, src


def run_video(n_frames, precision, warmup, dev, world, rank, batch):
    """configs[3]: frames [0, n_frames) of a 1080p video, generated on the device per batch from seed = frame
    index (no rank ever holds the video), contiguous frame ranges per rank, ONE all-gather of the scores."""
    wl = WORKLOADS["video10k"]
    H, W = wl["H"], wl["W"]
    net, prec, src = make_model("DISTS", precision, dev, H, W)

    def score_batch(lo, hi):
        ref, ren = video.synthetic_frames(range(lo, hi), H, W, dev)
        return net(ref, ren)

    with torch.no_grad():
        for _ in range(max(1, warmup)):
            score_batch(0, batch)
        if world > 1:
            sharding.gather_scores(torch.zeros(-(-n_frames // world), device=dev), n_frames)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        ops.timing_enable(True)
        t0 = time.perf_counter()
        scores = sharding.score_frames_sharded(score_batch, n_frames, batch, dev)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    ktimes = ops.timing_collect()
    ops.timing_enable(False)
    assert scores.numel() == n_frames and torch.isfinite(scores).all()
    cols = video.video_columns("DISTS", scores.cpu().numpy())
    return dt, ktimes, prec, H, W, src, {k: float(v) for k, v in cols.items()}


def self_launch(n, argv):
    """`python bench.py --gpus N` typed as is: run the N ranks under torch.distributed.run as a child process
    (never exec: nothing here has touched the GPU yet, and nothing will in this process), relay rank 0's JSON
    line on stdout (everything else the ranks print goes to stderr) and return the child's exit code."""
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    # a wall-clock limit on the whole job (NQA_BENCH_LAUNCH_TIMEOUT seconds, default 30 min): a hung rank must not block
    # the launcher for ever -- the child's process GROUP is terminated and the launcher exits non-zero
    limit = float(os.environ.get("NQA_BENCH_LAUNCH_TIMEOUT", "1800"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
    import signal
    import threading

    def expire():
        sys.stderr.write(f"bench.py: the {n}-rank job exceeded {limit:.0f} s; terminating its process group\n")
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)  # (start_new_session: the child leads its own group -- nothing else is in it)
            except ProcessLookupError:
                return
            time.sleep(5)
    timer = threading.Timer(limit, expire)
    timer.daemon = True
    timer.start()
    try:
        for line in proc.stdout:
            if line.startswith('{"metric"'):
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        rc = proc.wait()
    finally:
        timer.cancel()
    return rc if rc >= 0 else 124  # (killed by the limit: the conventional timeout code)


def run_stub(steps, warmup, world, rank):
    """`--workload stub` (tests only, never a bench number): the launch / timing / all-gather skeleton of
    run_workload on CPU tensors with a placeholder per-pair score, so the N > 1 path of THIS file can be
    exercised under gloo on a box without a GPU (tests/test_bench_launch.py).  No HIP kernel runs."""
    B = 8
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand(B, 3, 32, 32, generator=g)
    y = (x + 0.1 * torch.randn(B, 3, 32, 32, generator=g)).clamp_(0, 1)
    model = lambda a, b: (a - b).abs().mean((1, 2, 3))  # noqa: E731
    scores = torch.empty(steps * B)
    for _ in range(warmup):
        model(x, y)
    if world > 1:
        sharding.gather_scores(scores, world * scores.numel())
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        scores[k * B:(k + 1) * B] = model(x, y)
    all_scores = sharding.gather_scores(scores, world * scores.numel()) if world > 1 else scores
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert all_scores.numel() == world * steps * B and torch.isfinite(all_scores).all()
    return dt, B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["stub"], default="1080p",
                    help="1080p (default, BASELINE configs[2]) | 256 | adists1080p | adists256 | video10k")
    ap.add_argument("--precision", default=None, help="default: DISTS auto (calibrated f16 / f32m / f32s), A-DISTS f32s; "
                                                      "or one of f16, f32m, f32s, f32, bf16")
    ap.add_argument("--only", action="store_true", help="skip the companion workloads of the N=1 line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override pairs per GPU per step (experiments only)")
    ap.add_argument("--vgg", default=VGG, help="VGG-16 weights of the headline line: a checkpoint path or a stand-in spec "
                    "synth:<seed>[:<gain>] (profiling the default at another weight magnitude; the companions name their own)")
    ap.add_argument("--frames", type=int, default=10000, help="frames of the video10k workload")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python bench.py --gpus N`: become the launcher (no GPU call has been made in this process)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE={world})")
    stub = args.workload == "stub"
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
        # one rank per GPU; the modulo only matters when the N>1 path is rehearsed on a 1-GPU box
        # (NQA_DIST_BACKEND=gloo, several ranks sharing cuda:0)
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if stub else os.environ.get("NQA_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def rank_times(dt):
        """(max over ranks, per-rank list) of a rank's wall time (one all_gather_into_tensor, as for the scores)."""
        if world == 1:
            return dt, [dt]
        t = torch.tensor([dt], dtype=torch.float32, device=dev)
        allt = torch.empty(world, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allt, t)
        per = [float(v) for v in allt.cpu().tolist()]
        return max(per), per

    if stub:
        dt, B = run_stub(args.steps, args.warmup, world, rank)
        dt, per = rank_times(dt)
        if rank == 0:
            print(json.dumps({"metric": "stub frame-pairs/s (launch-path test only, no HIP kernel ran)",
                              "value": round(world * B * args.steps / dt, 2), "unit": "frame-pairs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": "stub"},
                              "per_rank_pairs_per_s": [round(B * args.steps / t, 2) for t in per]}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    traffic = load_traffic()
    wl = WORKLOADS[args.workload]

    if args.workload == "video10k":
        batch = args.batch or wl["B"]
        dt, ktimes, prec, H, W, src, cols = run_video(args.frames, args.precision, args.warmup, dev, world, rank, batch)
        dt, per = rank_times(dt)
        if rank == 0:
            roof, hbm, kms = rooflines(ktimes, H, W, batch, prec, traffic_for(traffic, f"1080p/{prec}"))
            # this rank's launches cover its own shard only: per-launch figures stay valid, per-step ones are per batch
            per_rank = -(-args.frames // world)
            nsteps = max(1, -(-per_rank // batch))
            out = {"metric": "DISTS frame-pairs/s", "value": round(args.frames / dt, 2), "unit": "frame-pairs/s",
                   "n_gpus": world, "steps": nsteps, "warmup": max(1, args.warmup),
                   "ms_per_step": round(dt / nsteps * 1e3, 4),
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": prec,
                   "data": "synthetic",
                   "config": {"workload": wl["name"], "frames": args.frames, "pairs_per_gpu_per_step": batch,
                              "height": H, "width": W, "vgg_weights": src,
                              "sharding": f"contiguous frame ranges over {world} ranks, one all-gather of scores"},
                   "per_rank_pairs_per_s": [round(-(-args.frames // world) / t, 2) for t in per],
                   "video_columns": cols, "roofline": roof, "kernel_ms_per_step": kms, "roofline_hbm": hbm}
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    dt, ktimes, prec, B, H, W, src = run_workload(args.workload, args.precision, args.steps, args.warmup, dev, world,
                                                  rank, args.batch, vgg=args.vgg)
    dt, per = rank_times(dt)
    out = None
    if rank == 0:
        roof, hbm, kms = rooflines(ktimes, H, W, B, prec, traffic_for(traffic, f"{args.workload}/{prec}"), steps=args.steps)
        if wl["metric"] == "A-DISTS":
            roof["note"] += ONE_STREAM_NOTE
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
                       "vgg_weights": src, "sharding": f"frames/{world} ranks, one all-gather of scores",
                       "precision": (args.precision or "auto (shipped default)") + " -> " + prec
                                    + ("" if args.precision else " on the gain-1.0 stand-in weights; the same default runs f32m on the "
                                       "ImageNet-magnitude set (gain 1.3) and f32s on the stress set (gain 1.6): see "
                                       "config.default_by_weight_set and workloads 1080p/auto@gain1.3, @gain1.6"),
                       "auto_calibration": dict(AUTO_REPORT) if not args.precision and wl["metric"] == "DISTS" else None,
                       "timing": "hipEvent pairs around every kernel launch are recorded inside the timed region"},
            "per_rank_pairs_per_s": [round(B * args.steps / t, 2) for t in per],
            "roofline": roof,
            "kernel_ms_per_step": kms,
            "roofline_hbm": hbm,
        }
    if world == 1 and not args.only and args.workload == "1080p" and not args.batch:
        comp = {}
        defaults = {"gain 1.0 (synth:1234, the headline)": {"mode": prec, "pairs_per_s": out["value"],
                                                            "conv_frac_of_2.5PF": roof["frac"]}}
        for entry in COMPANIONS:
            key, cprec = entry[0], entry[1]
            vgg = entry[2] if len(entry) > 2 else VGG
            if (key, cprec, vgg) == (args.workload, prec, VGG):
                continue  # that is the headline
            cdt, ckt, cp, cb, ch, cw, csrc = run_workload(key, cprec, args.steps, args.warmup, dev, 1, 0, vgg=vgg)
            croof, chbm, ckms = rooflines(ckt, ch, cw, cb, cp, traffic_for(traffic, f"{key}/{cp}"), steps=args.steps)
            if WORKLOADS[key]["metric"] == "A-DISTS":
                croof["note"] += ONE_STREAM_NOTE
            name = f"{key}/{cp}" if cprec is not None else f"{key}/auto@gain{vgg.rsplit(':', 1)[1]}"
            comp[name] = {"workload": WORKLOADS[key]["name"], "metric": WORKLOADS[key]["metric"] + " frame-pairs/s",
                          "value": round(cb * args.steps / cdt, 2), "unit": "frame-pairs/s",
                          "ms_per_step": round(cdt / args.steps * 1e3, 4), "dtype": cp,
                          "pairs_per_gpu_per_step": cb, "roofline": croof, "kernel_ms_per_step": ckms,
                          "roofline_hbm": chbm}
            if cprec is None:  # the shipped default on another pinned weight set: say what it calibrated to, and why
                comp[name]["vgg_weights"] = csrc
                comp[name]["precision"] = "auto (shipped default) -> " + cp
                comp[name]["auto_calibration"] = dict(AUTO_REPORT)
                defaults[f"gain {vgg.rsplit(':', 1)[1]} ({vgg})"] = {"mode": cp, "pairs_per_s": comp[name]["value"],
                                                                    "conv_frac_of_2.5PF": croof["frac"],
                                                                    "frac_of_issued_mfma": croof.get("frac_of_issued_mfma")}
            torch.cuda.empty_cache()
        out["workloads"] = comp
        # the three pinned weight sets side by side: what a user's checkpoint gets depends on how it grows activations
        # (gain 1.3 is the ImageNet magnitude); the headline `value` is the first of these
        out["config"]["default_by_weight_set"] = defaults
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, W, adists=wl["metric"] == "A-DISTS")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
