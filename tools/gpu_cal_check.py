import sys, torch
sys.path.insert(0, '/root/repo')
from nerf_qa_amd.DISTS_pytorch import DISTS
dev = torch.device("cuda:0")
for gain in (1.0, 1.3, 1.6):
    m = DISTS(vgg16_path=f"synth:1234:{gain}").to(dev).eval()
    r = m.calibrate(dev)
    print("gain", gain, "->", r["choice"], {k: {a: (round(b, 8) if isinstance(b, float) else b) for a, b in r[k].items()} for k in ("f16", "f16w", "f32m4", "f32m", "f32m2")}, flush=True)
