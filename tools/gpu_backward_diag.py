"""Who is right when the HIP image gradient and float32 autograd over the oracle differ?  Both against float64 autograd
(development aid, GPU box)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from nerf_qa_amd import synth  # noqa: E402
from nerf_qa_amd.DISTS_pytorch import DISTS  # noqa: E402
from oracle import dists_oracle as do  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
m = DISTS(vgg16_path="synth:1234").to(dev).eval()
convs32 = do.convs_from_numpy(synth.vgg16_weights(1234))
convs64 = [(w.double(), b.double()) for w, b in convs32]
alpha, beta = m.alpha.detach().cpu(), m.beta.detach().cpu()


def oracle_grad(xn, yn, dt):
    x, y = torch.from_numpy(xn).to(dt).requires_grad_(), torch.from_numpy(yn).to(dt).requires_grad_()
    convs = convs64 if dt == torch.float64 else convs32
    if dt == torch.float64:  # the oracle's helpers build float32 constants: redo the pyramid in double
        mean = torch.tensor(do.IMAGENET_MEAN, dtype=dt).view(1, -1, 1, 1)
        std = torch.tensor(do.IMAGENET_STD, dtype=dt).view(1, -1, 1, 1)

        def pyr(img):
            h = (img - mean) / std
            feats, li = [img], 0
            for s, nconv in enumerate(do.STAGE_CONVS):
                if s > 0:
                    c = h.shape[1]
                    filt = do.hanning_filter().double()[None, None].repeat(c, 1, 1, 1)
                    h = (F.conv2d(h ** 2, filt, stride=2, padding=1, groups=c) + 1e-12).sqrt()
                for _ in range(nconv):
                    w, b = convs[li]
                    h = F.relu(F.conv2d(h, w, b, padding=1))
                    li += 1
                feats.append(h)
            return feats
        f0, f1 = pyr(x), pyr(y)
    else:
        f0, f1 = do.vgg_pyramid(x, convs), do.vgg_pyramid(y, convs)
    s1, s2 = do.dists_stats(f0, f1)
    a, b = alpha.to(dt).reshape(-1), beta.to(dt).reshape(-1)
    w = a.sum() + b.sum()
    score = 1 - ((a / w) * s1).sum(1) - ((b / w) * s2).sum(1)
    score.sum().backward()
    return x.grad.double(), y.grad.double()


for (h, w) in ((40, 56), (96, 112), (160, 192)):
    xn, yn = synth.frame_batch([11, 12], h, w, ["blur", "noise10"])
    g64 = oracle_grad(xn, yn, torch.float64)
    g32 = oracle_grad(xn, yn, torch.float32)
    xd, yd = torch.from_numpy(xn).to(dev).requires_grad_(), torch.from_numpy(yn).to(dev).requires_grad_()
    m(xd, yd, require_grad=True).sum().backward()
    gh = (xd.grad.cpu().double(), yd.grad.cpu().double())
    for i, name in enumerate("xy"):
        ref = g64[i]
        r = ref.pow(2).mean().sqrt()
        e_h, e_32, e_h32 = (gh[i] - ref), (g32[i] - ref), (gh[i] - g32[i])
        print(f"{h}x{w} d/d{name}: rms err / rms  hip vs f64 {e_h.pow(2).mean().sqrt() / r:.2e}   f32 autograd vs f64 "
              f"{e_32.pow(2).mean().sqrt() / r:.2e}   hip vs f32 autograd {e_h32.pow(2).mean().sqrt() / r:.2e}   "
              f"max/max hip {e_h.abs().max() / ref.abs().max():.2e} f32 {e_32.abs().max() / ref.abs().max():.2e}", flush=True)
