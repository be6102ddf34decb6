"""Image gradients of DISTS: `DISTS.forward(x, y, require_grad=True)` (nerf_qa/DISTS_pytorch/DISTS_pt.py:105-108), and of
A-DISTS' default `as_loss=True` (nerf_qa/ADISTS/ADISTS.py:139-141) through `PyramidTaps`.

The reference gets them by running forward_once with autograd enabled.  Here the VALUES (S1, S2) come from the fused
HIP forward as always; when a gradient is asked for, `DistsSimilarities.backward` re-runs the pyramid layer by layer in
the float-precision mode (f32s: three-term split products) to have the activations, and walks it backwards with the
kernels of csrc/nqa_backward.hip: ReLU mask -> conv with the flipped, transposed weights -> L2-pool gradient -> ... ->
conv1_1's gradient onto the three image planes.  The per-channel statistics' gradient (DISTS_pt.py:130-141) is an affine
map of the two taps with per-(pair, channel) coefficients; those few numbers are evaluated in float64 on the device.
The alpha/beta weighted sum stays the torch expression of DISTS_pt.py, so autograd chains d(score)/d(S1, S2) into
this Function and reaches alpha and beta on its own.  No VGG weight receives a gradient (they are frozen, :51-52).
"""
from __future__ import annotations

import math

import torch

from . import ops

C1 = C2 = 1e-6  # DISTS_pt.py:115-116


def _stats_grad(fx, fy, g1, g2, dims):
    """d(sum g1*S1 + g2*S2)/d(fx), /d(fy) for one tap.  fx, fy: float maps with the pixel axes `dims`; g1, g2: (B, C)
    upstream gradients of that tap's S1 / S2 columns.

    With dx = fx - mean(fx), dy = fy - mean(fy):  g_fx = [g1 dS1/dmx + g2 (2 dS2/dv dx + dS2/dcov dy)] / N, and for a mild
    distortion (S2 near 1) the two S2 terms cancel to a small remainder.  So the maps are centred FIRST (elementwise, in
    float32 around a float64 mean) and every per-channel sum is accumulated in float64: forming the same expression from
    separately rounded float32 means loses the remainder -- the error then grows with sqrt(N) (measured 1.5e-3 at 40x56,
    1.7e-2 at 160x192 against float64 autograd; this form: float32-autograd level)."""
    n = 1
    for d in dims:
        n *= fx.shape[d]
    f64 = torch.float64
    mx, my = fx.sum(dim=dims, dtype=f64) / n, fy.sum(dim=dims, dtype=f64) / n
    shape = [fx.shape[0]] + [1] * (fx.dim() - 1)
    cdim = [d for d in range(1, fx.dim()) if d not in dims][0]
    shape[cdim] = fx.shape[cdim]
    dx, dy = fx - mx.float().reshape(shape), fy - my.float().reshape(shape)
    sx, sy = dx.sum(dim=dims, dtype=f64) / n, dy.sum(dim=dims, dtype=f64) / n  # what the float32 rounding of the means left
    vx = (dx * dx).sum(dim=dims, dtype=f64) / n - sx * sx
    vy = (dy * dy).sum(dim=dims, dtype=f64) / n - sy * sy
    cov = (dx * dy).sum(dim=dims, dtype=f64) / n - sx * sy
    g1, g2 = g1.double(), g2.double()
    n1, d1 = 2 * mx * my + C1, mx * mx + my * my + C1
    ds1_dmx, ds1_dmy = (2 * my * d1 - 2 * mx * n1) / (d1 * d1), (2 * mx * d1 - 2 * my * n1) / (d1 * d1)
    n2, d2 = 2 * cov + C2, vx + vy + C2
    ds2_dcov, ds2_dv = 2 / d2, -n2 / (d2 * d2)
    b = 2 * g2 * ds2_dv / n   # coefficient of the map's own centred values
    c = g2 * ds2_dcov / n     # coefficient of the other map's centred values
    ax = g1 * ds1_dmx / n - (b * sx + c * sy)
    ay = g1 * ds1_dmy / n - (b * sy + c * sx)
    ax, ay, b, c = (t.float().reshape(shape) for t in (ax, ay, b, c))
    gx = b * dx
    gx.add_(c * dy).add_(ax)
    gy = b * dy
    gy.add_(c * dx).add_(ay)
    return gx, gy


def _backward_blobs(module, dev):
    """Per conv layer 1..12 the data-gradient layer: W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx], packed once per weights."""
    key = module._weights_key(dev)
    cache = getattr(module, "_bwd_blobs", None)
    if cache is None or cache[0] != key:
        convs = module._conv_modules()
        blobs = {l: ops.pack_conv_split(convs[l].weight.detach().flip(2, 3).transpose(0, 1).contiguous()).to(dev)
                 for l in range(1, 13)}
        cache = (key, blobs, convs[0].weight.detach().float().contiguous().to(dev))
        module.__dict__["_bwd_blobs"] = cache
    return cache[1], cache[2]


@torch.no_grad()
def pyramid_keep(module, imgs):
    """The pyramid of `imgs` (n,3,H,W float) again, layer by layer in f32s, keeping every activation:
    (acts {layer: NHWC}, taps [5 float NHWC maps], pooled [4 split16 maps])."""
    prec = "f32s"
    packed = module._packed_weights(imgs.device, prec)
    acts = {0: ops.conv1_1(imgs, packed, prec)}  # split16 (n,H,W,64)
    taps, pooled = [], []
    inp = acts[0]
    for l in range(1, 13):
        out = ops.conv3x3_relu(inp, l, packed, prec)  # float for the tapped layers, split16 otherwise
        acts[l] = out
        inp = out
        if l in ops.TAP_LAYERS:
            taps.append(out)
            if l != 12:
                pooled.append(ops.l2pool(out, prec))  # split16
                inp = pooled[-1]
    return acts, taps, pooled


@torch.no_grad()
def pyramid_backward(module, acts, taps, pooled, g_taps):
    """d/d(image) (n,3,H,W) of a scalar whose gradients with respect to the five tapped maps are `g_taps` (float NHWC, as
    `taps`), through the activations pyramid_keep kept.

    The data-gradient convolutions take split16 (f16 hi + lo) operands, whose dynamic range is f16's -- and image
    gradients of a mean-type score are tiny (~1/N per pixel: 1e-8 at a few thousand pixels, where a half is already
    subnormal).  Everything here is LINEAR in g, so g is renormalised by an exact power of two before every layer
    (max |g| into [128, 256)) and the accumulated exponent taken out of the final image gradient."""
    blobs, w0 = _backward_blobs(module, taps[0].device)
    # A tapped map is a ReLU output: where it is 0 the gradient stops (torch's own rule, grad * (out > 0)).  Applied HERE,
    # before anything is renormalised: A-DISTS' F.normalize gives an exactly dead channel a gradient of the order of
    # 1 / eps = 1e12 times its upstream (5e8 measured on a 40 x 56 pair), which the mask discards -- but a renormalisation
    # by max |g| taken with it in would push every live gradient below the halves' range first.
    g_taps = [g * (t > 0) for g, t in zip(g_taps, taps)]

    def normalise(t, k_total):
        mx = float(t.abs().max())
        if mx > 0 and math.isfinite(mx):
            k = 7 - math.frexp(mx)[1] + 1  # mx * 2^k in [128, 256)
            t = t * (2.0 ** k)
            k_total += k
        return t, k_total

    g, K = normalise(g_taps[4], 0)
    for l in range(12, 0, -1):
        gm = ops.relu_mask_split16(g, acts[l], l not in ops.TAP_LAYERS)
        g = ops.conv3x3_split(gm, blobs[l], ops.CONV_CIN[l], relu=False)  # gradient w.r.t. the layer's input
        if l in (2, 4, 7, 10):  # the input was the L2-pool of the previous stage's tap
            s = ops.CONV_STAGE[l]
            gt = g_taps[s - 1] * (2.0 ** K)  # the tap's own gradient, in the running scale
            ops.l2pool_backward(taps[s - 1], pooled[s - 1], g, gt)  # gt += pool gradient
            g = gt
        g, K = normalise(g, K)
    gm = g * (ops.split16_decode(acts[0]) > 0)  # d(ReLU) of relu1_1, float
    return ops.conv1_1_backward(gm, w0) * (2.0 ** -K)  # (n,3,H,W), the input normalisation included


@torch.no_grad()
def dists_backward(module, x, y, g1, g2):
    """(dL/dx, dL/dy) given dL/dS1, dL/dS2 (each (B, 1475)) for the pairs (x, y), float32 (B,3,H,W) on the GPU."""
    b = x.shape[0]
    xy = torch.cat([x, y]).float().contiguous()
    acts, taps, pooled = pyramid_keep(module, xy)
    # ---- gradients of the statistics with respect to the six taps ----
    off = 3
    g_taps = []
    for k, t in enumerate(taps):
        c = t.shape[-1]
        gx, gy = _stats_grad(t[:b], t[b:], g1[:, off:off + c], g2[:, off:off + c], dims=(1, 2))
        g_taps.append(torch.cat([gx, gy]).contiguous())
        off += c
    gx0, gy0 = _stats_grad(x.float(), y.float(), g1[:, :3], g2[:, :3], dims=(2, 3))  # tap 0 = the raw image (NCHW)
    gimg = pyramid_backward(module, acts, taps, pooled, g_taps)
    return gimg[:b] + gx0, gimg[b:] + gy0


class PyramidTaps(torch.autograd.Function):
    """images (n,3,H,W) -> the five tapped maps relu1_2 .. relu5_3 as float NCHW tensors, differentiable in the images
    (ADISTS.forward(as_loss=True) runs forward_once WITH autograd, ADISTS.py:139-141; the frozen VGG weights get no
    gradient).  Values from the fused f32s forward; backward = pyramid_keep + pyramid_backward."""

    @staticmethod
    def forward(ctx, imgs, module):
        prec = "f32s"
        imgs = imgs.float().contiguous()
        taps = ops.vgg_pyramid(imgs, module._packed_weights(imgs.device, prec), prec, module._ws)
        ctx.module = module
        ctx.save_for_backward(imgs)
        return tuple(ops.nhwc_to_nchw_f32(t, prec) for t in taps)

    @staticmethod
    def backward(ctx, *grads):
        (imgs,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        acts, taps, pooled = pyramid_keep(ctx.module, imgs)
        g_taps = [(torch.zeros_like(t) if g is None else g.detach().float().permute(0, 2, 3, 1).contiguous())
                  for g, t in zip(grads, taps)]
        return pyramid_backward(ctx.module, acts, taps, pooled, g_taps), None


class DistsSimilarities(torch.autograd.Function):
    """(x, y) -> (S1, S2), each (B, 1475), differentiable in x and y."""

    @staticmethod
    def forward(ctx, x, y, module):
        prec = "f32s"  # gradients are formed in float precision; the values come from the same mode for consistency
        s1, s2 = ops.dists_forward(x, y, module._packed_weights(x.device, prec), prec, module._ws)
        ctx.module = module
        ctx.save_for_backward(x, y)
        return s1, s2

    @staticmethod
    def backward(ctx, g1, g2):
        x, y = ctx.saved_tensors
        gx, gy = dists_backward(ctx.module, x, y, g1.contiguous(), g2.contiguous())
        return (gx if ctx.needs_input_grad[0] else None), (gy if ctx.needs_input_grad[1] else None), None
