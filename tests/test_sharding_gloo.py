"""Frame sharding + the single score all-gather, on CPU with the gloo backend (world_size 2 and 4).

The GPU kernels are not involved: a stub scorer stands in for the model so that the
partitioning, padding and gather logic of nerf_qa_amd.sharding (what bench.py --gpus N and
video.score_video use under RCCL) is exercised where no GPU exists.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_qa_amd import sharding
    calls = []

    def score_batch(lo, hi):
        calls.append((lo, hi))
        return torch.arange(lo, hi, dtype=torch.float32) * 0.5 + 1.0

    full = sharding.score_frames_sharded(score_batch, n_frames, batch, torch.device("cpu"))
    q.put((rank, full.numpy(), calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,batch", [(2, 10, 4), (2, 7, 3), (2, 1, 8), (2, 33, 32),
                                                  (4, 10001, 8), (4, 3, 8)],
                         ids=["w2_10", "w2_7", "w2_1", "w2_33", "w4_10001_uneven", "w4_3_empty_ranks"])
def test_sharded_scores_gathered_on_every_rank(world, n_frames, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(n_frames, dtype=np.float32) * 0.5 + 1.0
    seen = []
    for rank, full, calls in results:
        assert np.array_equal(full, want), f"rank {rank} got {full}"
        seen += calls
    # every frame scored exactly once, in batches no longer than `batch`
    covered = sorted(i for lo, hi in seen for i in range(lo, hi))
    assert covered == list(range(n_frames))
    assert all(0 < hi - lo <= batch for lo, hi in seen)


def test_shard_range_partition():
    from nerf_qa_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 9, 10000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            assert max(b - a for a, b in spans) <= -(-n // world) if n else True


def test_gather_without_process_group():
    from nerf_qa_amd.sharding import gather_scores
    t = torch.arange(5, dtype=torch.float32)
    assert torch.equal(gather_scores(t, 5), t)


def _video_worker(rank, world, port, n_frames, batch, h, w, q):
    """configs[3]'s own frame source (video.synthetic_frames, seed = frame index) under the sharded loop."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_qa_amd import sharding, video
    cpu = torch.device("cpu")
    seen = []

    def score_batch(lo, hi):
        ref, ren = video.synthetic_frames(range(lo, hi), h, w, cpu)
        seen.append((lo, hi))
        return (ref - ren).abs().mean((1, 2, 3)) + ref[:, 0, 0, 0]  # any pure function of the two frames

    full = sharding.score_frames_sharded(score_batch, n_frames, batch, cpu)
    q.put((rank, full.numpy(), seen))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch", [(21, 4), (16, 8), (5, 8)], ids=["21x4_uneven", "16x8", "5x8_short_rank"])
def test_synthetic_video_sharded_world2(n_frames, batch):
    """Shard boundaries and per-frame seeds: two ranks generating their own frame ranges produce, after the one
    all-gather, exactly the vector a single process gets from frames 0..N-1 (BASELINE configs[3], prep.py:181-198)."""
    from nerf_qa_amd import video
    h, w = 16, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_video_worker, args=(r, 2, port, n_frames, batch, h, w, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, ren = video.synthetic_frames(range(n_frames), h, w, torch.device("cpu"))
    want = ((ref - ren).abs().mean((1, 2, 3)) + ref[:, 0, 0, 0]).numpy()
    assert len(set(np.round(want, 6))) == n_frames  # every frame is its own
    for rank, full, seen in results:
        assert np.array_equal(full, want), rank
        lo, hi = min(s[0] for s in seen) if seen else 0, max(s[1] for s in seen) if seen else 0
        per = -(-n_frames // 2)
        assert (lo, hi) == (min(n_frames, rank * per), min(n_frames, (rank + 1) * per)) or not seen
    cols = video.video_columns("DISTS", results[0][1])
    assert cols["DISTS"].dtype == np.float32 and abs(float(cols["DISTS"]) - float(np.mean(want))) < 1e-7


def _agree_worker(rank, world, port, choices, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_qa_amd import sharding

    class Model:  # what agree_precision touches of a DISTS module
        precision = "auto"

        def __init__(self, choice):
            self.report = {"choice": choice, "source": "measured"}
            self._agreed = {}
            self.calibrations = 0

        def _weights_key(self, device):
            return ("stub", str(device))

        def precision_for(self, h, w, device=None):
            if self.precision != "auto":
                return self.precision
            if h * w < 128 * 128:
                return "f32s"
            hit = self._agreed.get(3)
            return hit[1] if hit else self.calibrate(device, h, w)["choice"]

        def calibrate(self, device, h, w):
            self.calibrations += 1
            return self.report

    m = Model(choices[rank])
    got = sharding.agree_precision(m, 1080, 1920, torch.device("cpu"))
    ncal = m.calibrations
    small = sharding.agree_precision(m, 64, 64, torch.device("cpu"))
    named = Model(choices[rank])
    named.precision = "f16"
    q.put((rank, got, m.precision_for(1080, 1920), m._agreed_report, small,
           sharding.agree_precision(named, 1080, 1920, torch.device("cpu")), ncal))
    dist.destroy_process_group()


@pytest.mark.parametrize("choices", [("f16", "f32m"), ("f32s", "f16w"), ("f16", "f16"), ("f32m", "f16", "f32s")])
def test_rank0_calibrates_and_every_rank_adopts_its_verdict(choices):
    """bench.py --gpus N / the video harness: one precision mode per video.  Rank 0's calibration verdict is broadcast
    (round 4); the other ranks run NO calibration of their own, whatever theirs would have said."""
    world = len(choices)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, choices, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = choices[0]
    for rank, got, after, report, small, named, ncal in results:
        assert got == want and after == want and report["agreed_over_ranks"] == world and report["choice"] == want
        assert ncal == (2 if rank == 0 else 0), (rank, ncal)  # (rank 0: precision_for + the report; others: nothing)
        assert small == "f32s" and named == "f16"  # nothing to agree on: tiny frames, named precision
