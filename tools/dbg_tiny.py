import sys, torch, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ctypes as C, os
from nerf_qa_amd import _lib
if os.environ.get("NQA_LIB"):  # an older library: bind only the symbols it has
    h = C.CDLL(os.environ["NQA_LIB"])
    _lib._SIGNATURES = {k: v for k, v in _lib._SIGNATURES.items() if hasattr(h, k)}
from nerf_qa_amd import ops, synth
from oracle import dists_oracle
dev = torch.device("cuda:0")
np_convs = synth.vgg16_weights(1234)
convs = dists_oracle.convs_from_numpy(np_convs)
for (h, w, b) in ((2, 3, 1), (2, 3, 2), (1, 6, 2), (1, 7, 2), (1, 12, 2), (2, 6, 2), (3, 2, 2), (3, 3, 2), (3, 3, 3), (1, 1, 1), (1, 1, 2)):
    xn, yn = synth.frame_batch([1000 + h * 131 + w + i for i in range(b)], h, w)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    for prec in ("f32",):
        packed = ops.pack_vgg_weights(np_convs, prec).to(dev)
        taps = ops.vgg_pyramid(x.to(dev), packed, prec)
        ref = dists_oracle.vgg_pyramid(x, convs)[1:]
        errs = [(ops.nhwc_to_nchw_f32(t, prec).cpu() - r).abs().max().item() for t, r in zip(taps, ref)]
        s1, s2 = ops.dists_forward(x.to(dev), y.to(dev), packed, prec)
        f0, f1 = dists_oracle.vgg_pyramid(x, convs), dists_oracle.vgg_pyramid(y, convs)
        r1, r2 = dists_oracle.dists_stats(f0, f1)
        off = [0, 3, 67, 195, 451, 963, 1475]
        e1 = [(s1.cpu()[:, off[k]:off[k+1]] - r1[:, off[k]:off[k+1]]).abs().max().item() for k in range(6)]
        e2 = [(s2.cpu()[:, off[k]:off[k+1]] - r2[:, off[k]:off[k+1]]).abs().max().item() for k in range(6)]
        if e1[0] > 1e-4:
            print("   S1 stage0 got", s1.cpu()[:, :3].tolist(), "ref", r1[:, :3].tolist())
        print(h, w, b, prec, "tap errs", ["%.1e" % e for e in errs], "S1", ["%.1e" % e for e in e1], "S2", ["%.1e" % e for e in e2], flush=True)
