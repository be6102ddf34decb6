"""A-DISTS f16 / f32 vs the CPU oracle on structured frames (development aid)."""
import os; os.environ.setdefault("NQA_VGG16_WEIGHTS", "synth:1234")  # dev tool: stand-in weights, asked for explicitly
import sys
import warnings
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from nerf_qa_amd import synth  # noqa: E402
from nerf_qa_amd.ADISTS import ADISTS  # noqa: E402
from oracle import adists_oracle, dists_oracle  # noqa: E402
from test_gpu_fullsize import _frames  # noqa: E402
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    a16, a32, ab = ADISTS(precision="f16").to(dev).eval(), ADISTS(precision="f32").to(dev).eval(), ADISTS(precision="bf16").to(dev).eval()
convs = dists_oracle.convs_from_numpy(synth.vgg16_weights(1234))
for (b, h, w) in ((4, 256, 256), (4, 128, 160)):
    x, y = _frames(b, h, w, dev, 11)
    with torch.no_grad():
        s16, s32, sb = a16(x, y, as_loss=False).cpu(), a32(x, y, as_loss=False).cpu(), ab(x, y, as_loss=False).cpu()
    ref = adists_oracle.adists(x.cpu(), y.cpu(), convs)
    print(f"{h}x{w} oracle {ref.tolist()}")
    print("   f32-oracle", (s32 - ref).tolist())
    print("   f16-oracle", (s16 - ref).tolist())
    print("   bf16-oracle", (sb - ref).tolist())
