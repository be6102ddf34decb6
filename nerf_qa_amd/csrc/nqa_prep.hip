// Input preparation on the device: the decode-side steps the
// reference's harnesses run on the CPU between a decoded uint8 frame and the metric's float32
// NCHW input.  All of it is byte/float streaming work bound by HBM; nothing here touches LDS.
//
//   ToTensor                      uint8 HWC -> float32 CHW / 255          prep.py:89, data.py:80
//   F.interpolate(bilinear,       float32 NCHW resize, no antialias       prep.py:93-95, data.py:81-82,
//     align_corners=False)                                                 test2_prep.py:437
//   transforms.functional.resize  PIL Image.resize(BILINEAR) on uint8:    DISTS_pt.py:213-215,
//     of a PIL image              antialiased two-pass fixed-point filter  test2_prep.py:112,225
//
// The PIL filter is Pillow's ImagingResample for 8-bit images (third-party, poetry.lock pins
// pillow 10.2.0; restated from its published algorithm): per output index a window
// [xmin, xmin+n) of triangle weights computed in double, normalised, rounded to 22-bit fixed
// point; horizontal pass then vertical pass, each rounding to uint8.
#include "nqa_common.h"

namespace nqa {

// ---- ToTensor ---------------------------------------------------------------------------------
// roundtrip != 0 evaluates prep.py:89-91's expression (float/255 -> ToPILImage: mul(255).byte()
// truncates -> ToTensor); in float32 that is the identity on all 256 levels (tests/test_prep_oracle.py).
__global__ __launch_bounds__(256) void u8hwc_to_f32nchw_kernel(const uint8_t *__restrict__ in, int HW,
                                                               int roundtrip, float *__restrict__ out) {
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const uint8_t *p = in + ((size_t)n * HW + i) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float f = (float)p[c] / 255.0f;
    if (roundtrip) f = (float)(uint8_t)(f * 255.0f) / 255.0f;
    out[((size_t)n * 3 + c) * HW + i] = f;
  }
}

// ---- F.interpolate(mode='bilinear', align_corners=False) ----------------------------------------
// torch's upsample_bilinear2d: src = in/out * (dst + 0.5) - 0.5 clamped at 0, the two taps and
// their weights in float, rows combined as w0*(row0) + w1*(row1).
__device__ inline void torch_src(int dst, int in, int out, int &i0, int &i1, float &l0, float &l1) {
  // one fused multiply-add, as torch's CPU kernel evaluates it (a separately rounded product moves
  // src by an ulp, i.e. the tap weights by ~3e-5 at 1080p)
  const float scale = (float)in / (float)out;
  float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void resize_bilinear_f32_kernel(const float *__restrict__ in, int Hin, int Win,
                                                                  int Hout, int Wout, float *__restrict__ out) {
#pragma clang fp contract(off)  // plain float products and sums, so the fused kernel below matches bit for bit
  const int plane = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Hout * Wout) return;
  const int y = i / Wout, x = i - y * Wout;
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  torch_src(y, Hin, Hout, y0, y1, ly0, ly1);
  torch_src(x, Win, Wout, x0, x1, lx0, lx1);
  const float *p = in + (size_t)plane * Hin * Win;
  out[(size_t)plane * Hout * Wout + i] = ly0 * (lx0 * p[y0 * Win + x0] + lx1 * p[y0 * Win + x1]) +
                                         ly1 * (lx0 * p[y1 * Win + x0] + lx1 * p[y1 * Win + x1]);
}

// ToTensor + F.interpolate in one pass: only the four taps of every output pixel are read, and the
// full-resolution float tensor (12 bytes per source pixel, written then re-read) never exists.
// Each tap is converted as ToTensor would ((float)v / 255), so the result is bit-identical to
// running the two kernels above back to back.
__global__ __launch_bounds__(256) void u8_resize_bilinear_kernel(const uint8_t *__restrict__ in, int Hin, int Win,
                                                                 int Hout, int Wout, float *__restrict__ out) {
#pragma clang fp contract(off)
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Hout * Wout) return;
  const int y = i / Wout, x = i - y * Wout;
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  torch_src(y, Hin, Hout, y0, y1, ly0, ly1);
  torch_src(x, Win, Wout, x0, x1, lx0, lx1);
  const uint8_t *p = in + (size_t)n * Hin * Win * 3;
  const uint8_t *p00 = p + ((size_t)y0 * Win + x0) * 3, *p01 = p + ((size_t)y0 * Win + x1) * 3;
  const uint8_t *p10 = p + ((size_t)y1 * Win + x0) * 3, *p11 = p + ((size_t)y1 * Win + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = (float)p00[c] / 255.0f, b = (float)p01[c] / 255.0f;
    const float cc = (float)p10[c] / 255.0f, d = (float)p11[c] / 255.0f;
    out[((size_t)n * 3 + c) * Hout * Wout + i] = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);
  }
}

// ---- split16 <-> float (NQA_PREC_F32S activation format, nqa_common.h) ---------------------------
__global__ __launch_bounds__(256) void split16_encode_kernel(const float *__restrict__ in, long groups, int C,
                                                             char *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one 4-channel group
  if (i >= groups) return;
  const int G = C / 4;
  const long pix = i / G;
  const int c = (int)(i - pix * G) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4 *>(in + pix * C + c);
  store_split4(out + pix * C * 4, c, v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void split16_decode_kernel(const char *__restrict__ in, long groups, int C,
                                                             float *__restrict__ out) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups) return;
  const int G = C / 4;
  const long pix = i / G;
  const int c = (int)(i - pix * G) * 4;
  const char *p = in + pix * C * 4 + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
  const h4 hi = *reinterpret_cast<const h4 *>(p), lo = *reinterpret_cast<const h4 *>(p + 32);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)hi[e] + (float)lo[e];
  *reinterpret_cast<f32x4 *>(out + pix * C + c) = v;
}

// ---- PIL Image.resize(BILINEAR), uint8 ----------------------------------------------------------
static constexpr int kPrecBits = 32 - 8 - 2;

struct PilAxis {
  int in, out, ksize;
};
static PilAxis pil_axis(int in, int out) {
  double fs = (double)in / out;
  if (fs < 1.0) fs = 1.0;
  return {in, out, (int)ceil(1.0 * fs) * 2 + 1};
}

// bounds[2*xx] = xmin, bounds[2*xx+1] = count, kk[xx*ksize + x] = fixed-point weight.
// One thread per output index; all arithmetic in double with contraction off, as compiled C.
__global__ __launch_bounds__(64) void pil_coeffs_kernel(int in, int out, int ksize, int *__restrict__ bounds,
                                                        int *__restrict__ kk) {
#pragma clang fp contract(off)
  const int xx = blockIdx.x * 64 + threadIdx.x;
  if (xx >= out) return;
  const double scale = (double)in / (double)out;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in) xmax = in;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double t = ((double)(x + xmin) - center + 0.5) * ss;
    if (t < 0.0) t = -t;
    ww += t < 1.0 ? 1.0 - t : 0.0;
  }
  int *k = kk + (size_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double t = ((double)(x + xmin) - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      w = t < 1.0 ? 1.0 - t : 0.0;
      if (ww != 0.0) w /= ww;
    }
    k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << kPrecBits)) : (int)(0.5 + w * (double)(1 << kPrecBits));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ inline uint8_t clip8(int v) {
  v >>= kPrecBits;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One pass along an axis of a (n, A, Bdim, 3) uint8 tensor.  HORIZ: the filtered axis is the inner
// pixel axis (stride 3 bytes), else the row axis (stride row_bytes).
template <bool HORIZ>
__global__ __launch_bounds__(256) void pil_pass_kernel(const uint8_t *__restrict__ in, int Hin, int Win, int Hout,
                                                       int Wout, int ksize, const int *__restrict__ bounds,
                                                       const int *__restrict__ kk, uint8_t *__restrict__ out) {
  const int n = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Hout * Wout) return;
  const int y = i / Wout, x = i - y * Wout;
  const int o = HORIZ ? x : y;
  const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
  const int *k = kk + (size_t)o * ksize;
  const uint8_t *p = in + (size_t)n * Hin * Win * 3 + (HORIZ ? ((size_t)y * Win + lo) * 3 : ((size_t)lo * Win + x) * 3);
  const size_t step = HORIZ ? 3 : (size_t)Win * 3;
  int s0 = 1 << (kPrecBits - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < cnt; ++t) {
    const int w = k[t];
    s0 += (int)p[0] * w;
    s1 += (int)p[1] * w;
    s2 += (int)p[2] * w;
    p += step;
  }
  uint8_t *q = out + ((size_t)n * Hout * Wout + i) * 3;
  q[0] = clip8(s0);
  q[1] = clip8(s1);
  q[2] = clip8(s2);
}

}  // namespace nqa

using namespace nqa;

static bool bad_img(const char *who, const void *a, const void *b, int n, int h, int w) {
  if (!a || !b) {
    set_error("%s: null pointer", who);
    return true;
  }
  if (n <= 0 || n > 65535 || h <= 0 || w <= 0 || (long)h * w >= (1L << 29)) {  // n rides in gridDim.y
    set_error("%s: bad size n=%d H=%d W=%d", who, n, h, w);
    return true;
  }
  return false;
}

extern "C" {

int nqa_u8hwc_to_f32nchw(const uint8_t *in, int n, int H, int W, int pil_roundtrip, float *out, void *stream) {
  if (bad_img("u8hwc_to_f32nchw", in, out, n, H, W)) return NQA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TimedLaunch t(NQA_K_PREP, st);
  u8hwc_to_f32nchw_kernel<<<dim3(cdiv(H * W, 256), n), 256, 0, st>>>(in, H * W, pil_roundtrip, out);
  return check_launch("u8hwc_to_f32nchw");
}

int nqa_resize_bilinear_f32(const float *in, int planes, int Hin, int Win, int Hout, int Wout, float *out,
                            void *stream) {
  if (bad_img("resize_bilinear_f32", in, out, planes, Hin, Win) ||
      bad_img("resize_bilinear_f32", in, out, planes, Hout, Wout))
    return NQA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TimedLaunch t(NQA_K_PREP, st);
  resize_bilinear_f32_kernel<<<dim3(cdiv(Hout * Wout, 256), planes), 256, 0, st>>>(in, Hin, Win, Hout, Wout, out);
  return check_launch("resize_bilinear_f32");
}

int nqa_u8_resize_bilinear_f32(const uint8_t *in, int n, int Hin, int Win, int Hout, int Wout, float *out,
                               void *stream) {
  if (bad_img("u8_resize_bilinear_f32", in, out, n, Hin, Win) ||
      bad_img("u8_resize_bilinear_f32", in, out, n, Hout, Wout))
    return NQA_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  TimedLaunch t(NQA_K_PREP, st);
  u8_resize_bilinear_kernel<<<dim3(cdiv(Hout * Wout, 256), n), 256, 0, st>>>(in, Hin, Win, Hout, Wout, out);
  return check_launch("u8_resize_bilinear_f32");
}

static int split16_args(const char *who, const void *in, const void *out, long pixels, int C) {
  if (!in || !out) {
    set_error("%s: null pointer", who);
    return NQA_E_ARG;
  }
  if (pixels <= 0 || C <= 0 || C % 16 || pixels * (C / 4) / 256 >= (1L << 31)) {
    set_error("%s: bad size pixels=%ld C=%d (C must be a multiple of 16)", who, pixels, C);
    return NQA_E_ARG;
  }
  return NQA_OK;
}

int nqa_split16_encode(const float *in_nhwc, long pixels, int C, void *out, void *stream) {
  int rc = split16_args("split16_encode", in_nhwc, out, pixels, C);
  if (rc) return rc;
  const long groups = pixels * (C / 4);
  split16_encode_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      in_nhwc, groups, C, static_cast<char *>(out));
  return check_launch("split16_encode");
}

int nqa_split16_decode(const void *in, long pixels, int C, float *out_nhwc, void *stream) {
  int rc = split16_args("split16_decode", in, out_nhwc, pixels, C);
  if (rc) return rc;
  const long groups = pixels * (C / 4);
  split16_decode_kernel<<<(unsigned)((groups + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<const char *>(in), groups, C, out_nhwc);
  return check_launch("split16_decode");
}

size_t nqa_resize_pil_workspace_bytes(int n, int Hin, int Win, int Hout, int Wout) {
  if (n <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return 0;
  const PilAxis ax = pil_axis(Win, Wout), ay = pil_axis(Hin, Hout);
  return align_up((size_t)n * Hin * Wout * 3, 256) + align_up((size_t)Wout * (ax.ksize + 2) * 4, 256) +
         align_up((size_t)Hout * (ay.ksize + 2) * 4, 256);
}

int nqa_resize_pil_bilinear_u8(const uint8_t *in, int n, int Hin, int Win, int Hout, int Wout, void *workspace,
                               size_t workspace_bytes, uint8_t *out, void *stream) {
  if (bad_img("resize_pil_bilinear_u8", in, out, n, Hin, Win) ||
      bad_img("resize_pil_bilinear_u8", in, out, n, Hout, Wout))
    return NQA_E_ARG;
  if (!workspace || workspace_bytes < nqa_resize_pil_workspace_bytes(n, Hin, Win, Hout, Wout)) {
    set_error("resize_pil_bilinear_u8: workspace %zu < %zu bytes", workspace_bytes,
              nqa_resize_pil_workspace_bytes(n, Hin, Win, Hout, Wout));
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const PilAxis ax = pil_axis(Win, Wout), ay = pil_axis(Hin, Hout);
  char *base = static_cast<char *>(workspace);
  uint8_t *tmp = reinterpret_cast<uint8_t *>(base);
  int *bx = reinterpret_cast<int *>(base + align_up((size_t)n * Hin * Wout * 3, 256));
  int *kx = bx + 2 * Wout;
  int *by = reinterpret_cast<int *>(reinterpret_cast<char *>(bx) + align_up((size_t)Wout * (ax.ksize + 2) * 4, 256));
  int *ky = by + 2 * Hout;
  TimedLaunch t(NQA_K_PREP, st);
  // Pillow skips a pass whose size does not change (ImagingResampleInner: need_horizontal /
  // need_vertical); an identity pass would be exact anyway, so it is only skipped for speed.
  const bool need_h = Wout != Win, need_v = Hout != Hin;
  const uint8_t *src = in;
  if (need_h) {
    pil_coeffs_kernel<<<cdiv(Wout, 64), 64, 0, st>>>(Win, Wout, ax.ksize, bx, kx);
    pil_pass_kernel<true><<<dim3(cdiv(Hin * Wout, 256), n), 256, 0, st>>>(in, Hin, Win, Hin, Wout, ax.ksize, bx, kx,
                                                                          need_v ? tmp : out);
    src = tmp;
  }
  if (need_v) {
    pil_coeffs_kernel<<<cdiv(Hout, 64), 64, 0, st>>>(Hin, Hout, ay.ksize, by, ky);
    pil_pass_kernel<false><<<dim3(cdiv(Hout * Wout, 256), n), 256, 0, st>>>(src, Hin, Wout, Hout, Wout, ay.ksize, by,
                                                                            ky, out);
  }
  if (!need_h && !need_v &&
      hipMemcpyAsync(out, in, (size_t)n * Hin * Win * 3, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_error("resize_pil_bilinear_u8: copy failed");
    return NQA_E_LAUNCH;
  }
  return check_launch("resize_pil_bilinear_u8");
}

}  // extern "C"
