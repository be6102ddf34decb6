"""The input-preparation oracle (oracle/prep_oracle.py) against the real third-party code: Pillow's
Image.resize(BILINEAR) bit for bit, and torch's own ToTensor / F.interpolate expressions."""
import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def _img(seed, h, w):
    from nerf_qa_amd import synth
    return (synth.uniform(seed, h * w * 3) * 256).astype(np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("hin,win,hout,wout", [
    (300, 400, 256, 256), (300, 400, 256, 341), (1080, 1920, 256, 455), (37, 53, 64, 80), (90, 61, 45, 61),
    (257, 258, 256, 256), (64, 64, 64, 32), (513, 700, 256, 349), (40, 30, 41, 29)])
def test_pil_restatement_is_bit_exact(hin, win, hout, wout):
    from oracle import prep_oracle
    img = _img(hin * 7 + win, hin, win)
    want = np.asarray(Image.fromarray(img).resize((wout, hout), Image.BILINEAR))
    got = prep_oracle.pil_resize_bilinear_u8(img, (hout, wout))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_smooth_image_and_extremes():
    from oracle import prep_oracle
    yy, xx = np.mgrid[0:333, 0:517]
    img = np.stack([(yy * 255 // 332), (xx * 255 // 516), np.where((yy + xx) % 2 == 0, 255, 0)], -1).astype(np.uint8)
    want = np.asarray(Image.fromarray(img).resize((256, 256), Image.BILINEAR))
    assert np.array_equal(prep_oracle.pil_resize_bilinear_u8(img, (256, 256)), want)


def test_size_policies():
    from nerf_qa_amd import prep
    assert prep.pil_resize_size(1080, 1920, False) == (256, 256)
    assert prep.pil_resize_size(1080, 1920, True) == (256, 455)
    assert prep.pil_resize_size(1920, 1080, True) == (455, 256)
    assert prep.pil_resize_size(200, 900, True) == (200, 900)  # short side <= 256: untouched (DISTS_pt.py:211)
    h, w = prep.equal_pixel_size(1080, 1920)
    assert (h, w) == (192, 341) and abs(h * w - 65536) < 400
    assert prep.equal_pixel_size(1920, 1080) == (341, 192)


def test_to_tensor_roundtrip_is_identity():
    """prep.py:89-91: (v/255)*255 truncated back to a byte -- in float32 this returns v for all 256 levels,
    so the ToPILImage -> ToTensor round trip changes nothing (the device kernel still mirrors the expression)."""
    from oracle import prep_oracle
    v = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, 3)
    assert torch.equal(prep_oracle.to_tensor(v), prep_oracle.to_tensor_roundtrip(v))
