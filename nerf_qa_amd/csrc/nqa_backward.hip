// Backward pass of the DISTS pyramid for gfx950: what DISTS.forward(x, y, require_grad=True) needs beyond the forward
// kernels (nerf_qa/DISTS_pytorch/DISTS_pt.py:105-108: the reference simply runs forward_once WITH autograd).
//
// The chain, per conv layer from relu5_3 down (driven by nerf_qa_amd/autograd.py, which re-runs the forward layer by
// layer to have the activations -- the fast fused forward keeps none):
//   relu_mask_split16   g * (act > 0) as split16 records            d(ReLU), and the input format of the next conv
//   conv3x3_split_generic (nqa_conv.hip)                             d(conv)/d(input) = conv with the flipped, transposed
//                                                                    weights, three-term split products (float accuracy)
//   l2pool_backward     g_tap += x * sum_o w(o) g_o / y_o            d(sqrt(hanning3x3_s2(x^2) + 1e-12)), DISTS_pt.py:22-25
//   conv1_1_backward    64 channels -> the 3 image planes            d(conv1_1), NCHW float out
// None of this is on the scoring hot path (no full-reference caller of the reference asks for image gradients); the
// kernels are plain HBM-streaming code.
#include "nqa_common.h"

namespace nqa {

__device__ static inline float split16_value(const char *pixel, int c) {  // channel c of a split16 pixel record
  const char *p = pixel + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
  return (float)*reinterpret_cast<const _Float16 *>(p) + (float)*reinterpret_cast<const _Float16 *>(p + 32);
}

// out = split16(g * (act > 0)); one thread per (pixel, 4 channels).  act is float (a tapped map) or split16.
__global__ __launch_bounds__(256) void relu_mask_split16_kernel(const float *__restrict__ g, const void *__restrict__ act,
                                                                int act_split, long npix, int C, char *__restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int G = C / 4;
  if (idx >= npix * G) return;
  const long p = idx / G;
  const int c = (int)(idx - p * G) * 4;
  const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + p * C + c);
  float a[4];
  if (act_split) {
    const char *rec = static_cast<const char *>(act) + p * (long)C * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = split16_value(rec, c + e);
  } else {
    const f32x4 av = *reinterpret_cast<const f32x4 *>(static_cast<const float *>(act) + p * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = av[e];
  }
  store_split4(out + p * (long)C * 4, c, a[0] > 0.f ? gv[0] : 0.f, a[1] > 0.f ? gv[1] : 0.f, a[2] > 0.f ? gv[2] : 0.f,
               a[3] > 0.f ? gv[3] : 0.f);
}

// g_x[iy, ix, c] += x[iy, ix, c] * sum over the (up to four) pooled pixels o whose 3x3 window holds (iy, ix) of
// w(o; iy, ix) * g_y[o, c] / y[o, c],  w = hanning3x3 / 16, y = the pooled map (split16 records, as the forward wrote it)
__global__ __launch_bounds__(256) void l2pool_backward_kernel(const float *__restrict__ x, const char *__restrict__ y,
                                                              const float *__restrict__ gy, int H, int W, int C, int Ho,
                                                              int Wo, long total, float *__restrict__ gx) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int G = C / 4;
  const int c = (int)(idx % G) * 4;
  long t = idx / G;
  const int ix = (int)(t % W);
  t /= W;
  const int iy = (int)(t % H);
  const int n = (int)(t / H);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // output rows oy with |2 oy - iy| <= 1: iy even -> oy = iy/2 (centre row, weight 1/2); iy odd -> oy = (iy-1)/2 and
  // (iy+1)/2 (edge rows, weight 1/4 each)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = (iy & 1) ? (iy - 1) / 2 + a : (a == 0 ? iy / 2 : -1);
    if (oy < 0 || oy >= Ho) continue;
    const float wy = (iy & 1) ? 0.25f : 0.5f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ox = (ix & 1) ? (ix - 1) / 2 + b : (b == 0 ? ix / 2 : -1);
      if (ox < 0 || ox >= Wo) continue;
      const float wgt = wy * ((ix & 1) ? 0.25f : 0.5f);
      const long o = ((long)n * Ho + oy) * Wo + ox;
      const f32x4 gv = *reinterpret_cast<const f32x4 *>(gy + o * C + c);
      const char *rec = y + o * (long)C * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += wgt * gv[e] / split16_value(rec, c + e);
    }
  }
  const long xi = (((long)n * H + iy) * W + ix) * C + c;
  const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + xi);
  f32x4 o = *reinterpret_cast<const f32x4 *>(gx + xi);
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] += xv[e] * acc[e];
  *reinterpret_cast<f32x4 *>(gx + xi) = o;
}

// d(conv1_1)/d(normalised image): gimg[n, c, y, x] = sum_{ky,kx,co} gm[n, y+1-ky, x+1-kx, co] * w[co, c, ky, kx], then
// divided by std[c] (the input normalisation (x - mean) / std, DISTS_pt.py:92).  gm = g * (relu1_1 > 0), float NHWC.
// One wave per output pixel: lane = output channel co of gm, 27 products per lane, wave sums.
__global__ __launch_bounds__(256) void conv1_1_backward_kernel(const float *__restrict__ gm, const float *__restrict__ w,
                                                               int H, int W, long npix, float *__restrict__ gimg) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const long HW = (long)H * W;
  const int n = (int)(pix / HW);
  const long r = pix - (long)n * HW;
  const int y = (int)(r / W), x = (int)(r - (long)y * W);
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int sy = y + 1 - ky;
    if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int sx = x + 1 - kx;
      if ((unsigned)sx >= (unsigned)W) continue;
      const float gv = gm[(((long)n * H + sy) * W + sx) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = fmaf(gv, w[((lane * 3 + c) * 3 + ky) * 3 + kx], acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_down(acc[c], off, 64);
  if (lane == 0) {
    const float sd[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) gimg[((long)n * 3 + c) * HW + r] = acc[c] / sd[c];
  }
}

int relu_mask_split16(const float *g, const void *act, int act_split, long npix, int C, void *out, hipStream_t st) {
  const long total = npix * (C / 4);
  relu_mask_split16_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(g, act, act_split, npix, C,
                                                                                  static_cast<char *>(out));
  return check_launch("relu_mask_split16");
}

int l2pool_backward(const float *x, const void *y_split16, const float *gy, int n, int H, int W, int C, float *gx,
                    hipStream_t st) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)n * H * W * (C / 4);
  l2pool_backward_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(x, static_cast<const char *>(y_split16), gy,
                                                                                H, W, C, Ho, Wo, total, gx);
  return check_launch("l2pool_backward");
}

int conv1_1_backward(const float *gm, const float *w_oihw, int n, int H, int W, float *gimg, hipStream_t st) {
  const long npix = (long)n * H * W;
  conv1_1_backward_kernel<<<dim3((unsigned)((npix + 3) / 4)), 256, 0, st>>>(gm, w_oihw, H, W, npix, gimg);
  return check_launch("conv1_1_backward");
}

}  // namespace nqa
