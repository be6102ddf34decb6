"""BASELINE.json configs[3] on the GPU: the synthetic video (frames generated on the device from seed = frame
index) scored through sharding.score_frames_sharded exactly as bench.py --workload video10k does, against direct
module calls at 1080p and against the CPU oracle at 64x96 (prep.py:181-198 is the loop being replaced)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_video_1080p_three_batches_equal_direct_calls(dev):
    from nerf_qa_amd import sharding, video
    from nerf_qa_amd.DISTS_pytorch import DISTS
    H, W, B, N = 1080, 1920, 8, 21  # two full batches and a ragged third
    net = DISTS(vgg16_path="synth:1234").to(dev).eval()
    calls = []

    def score_batch(lo, hi):
        calls.append((lo, hi))
        ref, ren = video.synthetic_frames(range(lo, hi), H, W, dev)
        return net(ref, ren)

    with torch.no_grad():
        scores = sharding.score_frames_sharded(score_batch, N, B, dev)
        assert calls == [(0, 8), (8, 16), (16, 21)] and scores.shape == (N,) and scores.dtype == torch.float32
        # the same frames, scored one direct call per frame range of a different batching
        direct = []
        for lo, hi in ((0, 5), (5, 13), (13, 21)):
            ref, ren = video.synthetic_frames(range(lo, hi), H, W, dev)
            direct.append(net(ref, ren))
        direct = torch.cat(direct)
    # a pair's score does not depend on its batch neighbours.  The statistics' block partition follows the batch size,
    # so with float taps (f32s, the mixed modes' deep stages) the float32 partial sums of a channel are grouped
    # differently -- 1e-7 (in every mode since round 4: the fused kernels of taps 1-2 sum in float32 per block)
    assert (scores - direct).abs().max().item() <= 3e-7, (scores - direct).abs().max().item()
    net16 = DISTS(vgg16_path="synth:1234", precision="f16").to(dev).eval()
    with torch.no_grad():
        a = torch.cat([net16(*video.synthetic_frames(range(lo, hi), H, W, dev)) for lo, hi in ((0, 8), (8, 13))])
        b2 = torch.cat([net16(*video.synthetic_frames(range(lo, hi), H, W, dev)) for lo, hi in ((0, 5), (5, 13))])
        a2 = torch.cat([net16(*video.synthetic_frames(range(lo, hi), H, W, dev)) for lo, hi in ((0, 8), (8, 13))])
    # the same calls again are bit-equal (fixed block -> tile partition, no atomics); a different split of the frames into
    # batches moves the block boundaries of the fused conv + statistics kernels' float32 partial sums: 1e-7, as above
    assert torch.equal(a, a2)
    assert (a - b2).abs().max().item() <= 3e-7, (a - b2).abs().max().item()
    del net16
    assert torch.isfinite(scores).all() and scores.min() > 0 and len(set(scores.cpu().tolist())) == N
    cols = video.video_columns("DISTS", scores.cpu().numpy())
    assert cols["DISTS"].dtype == np.float32 and cols["DISTS_min"] <= cols["DISTS"] <= cols["DISTS_max"]
    del net
    torch.cuda.empty_cache()


def test_video_64x96_against_the_oracle(dev):
    from nerf_qa_amd import sharding, synth, video
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch import DISTS
    from oracle import adists_oracle, dists_oracle
    H, W, B, N = 64, 96, 8, 19
    net = DISTS(vgg16_path="synth:1234").to(dev).eval()
    anet = ADISTS(vgg16_path="synth:1234").to(dev).eval()
    with torch.no_grad():
        got = sharding.score_frames_sharded(
            lambda lo, hi: net(*video.synthetic_frames(range(lo, hi), H, W, dev)), N, B, dev).cpu()
        agot = sharding.score_frames_sharded(
            lambda lo, hi: anet(*video.synthetic_frames(range(lo, hi), H, W, dev), as_loss=False), N, B, dev).cpu()
    ref, ren = (t.cpu() for t in video.synthetic_frames(range(N), H, W, dev))  # the device's frames, judged on CPU
    convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234))
    want = dists_oracle.dists(ref, ren, convs, net.alpha.detach().cpu(), net.beta.detach().cpu())
    awant = adists_oracle.adists(ref, ren, convs)
    err, aerr = (got - want).abs().max().item(), (agot - awant).abs().max().item()
    print(f"\nvideo 64x96 x{N}: DISTS max|d|={err:.2e}  A-DISTS max|d|={aerr:.2e}")
    assert err <= 1e-4 and aerr <= 1e-4
