"""The C ABI from several host threads at once (include/nqa.h: state is per thread -- error string, timing ring,
kernel-variant switches): two threads, each with its own stream, module and workspace, score different batches
concurrently; every score must equal, bit for bit, what the same call returns when it runs alone."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batches(dev, n, h, w):
    g = torch.Generator(device=dev)
    out = []
    for i in range(n):
        g.manual_seed(1000 + i)
        x = torch.rand(2, 3, h, w, device=dev, generator=g)
        y = (x + 0.05 * torch.randn(2, 3, h, w, device=dev, generator=g)).clamp(0, 1)
        out.append((x, y))
    return out


@pytest.mark.parametrize("metric", ["DISTS", "ADISTS"])
def test_two_threads_score_concurrently(metric):
    from nerf_qa_amd import _lib
    from nerf_qa_amd.ADISTS import ADISTS
    from nerf_qa_amd.DISTS_pytorch import DISTS
    dev = torch.device("cuda:0")
    data = _batches(dev, 6, 160, 224)
    make = (lambda: DISTS().to(dev).eval()) if metric == "DISTS" else (lambda: ADISTS().to(dev).eval())
    call = (lambda m, x, y: m(x, y, batch_average=False)) if metric == "DISTS" else (lambda m, x, y: m(x, y, as_loss=False))
    ref_model = make()
    with torch.no_grad():
        alone = [call(ref_model, x, y).clone() for x, y in data]
    torch.cuda.synchronize(dev)

    results, errors = {}, []
    start = threading.Barrier(2)

    def worker(tid):
        try:
            model, stream = make(), torch.cuda.Stream(device=dev)
            stream.wait_stream(torch.cuda.default_stream(dev))
            start.wait()
            with torch.no_grad(), torch.cuda.stream(stream):
                for rep in range(4):
                    for i in range(tid, len(data), 2):
                        results[(tid, rep, i)] = call(model, *data[i]).clone()
                # an argument error on this thread must not leak into the other thread's error string
                rc = _lib.lib().nqa_dists_score(None, None, None, None, -1, None, None)
                assert rc != 0 and _lib.lib().nqa_last_error()
            stream.synchronize()
        except Exception as e:  # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 4 * len(data)
    for (tid, rep, i), s in results.items():
        assert torch.equal(s, alone[i]), (metric, tid, rep, i, (s - alone[i]).abs().max().item())


def test_timing_ring_of_a_worker_thread_is_released_at_thread_exit():
    """The per-thread timing ring (hipEvent pairs, created lazily) is given back when the thread ends; the main
    thread's calls before and after are undisturbed and untimed."""
    import threading
    from nerf_qa_amd import ops, synth
    from nerf_qa_amd.DISTS_pytorch import DISTS
    dev = torch.device("cuda:0")
    m = DISTS(precision="f32s", vgg16_path="synth:1234").to(dev).eval()
    xn, yn = synth.frame_batch([1, 2], 48, 64)
    x, y = torch.from_numpy(xn).to(dev), torch.from_numpy(yn).to(dev)
    with torch.no_grad():
        want = m(x, y).clone()
    seen = {}

    def worker():
        ops.timing_enable(True)
        with torch.no_grad():
            got = m(x, y)
        torch.cuda.synchronize(dev)
        seen["kt"] = ops.timing_collect()
        seen["equal"] = bool(torch.equal(got, want))

    for _ in range(3):  # three generations of threads, each with its own ring
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert seen["equal"] and seen["kt"]["conv_igemm"][0] == 12 and seen["kt"]["conv_igemm"][1] > 0
    kt = ops.timing_collect()  # this thread never enabled timing
    assert all(v[0] == 0 for v in kt.values())
    with torch.no_grad():
        assert torch.equal(m(x, y), want)
