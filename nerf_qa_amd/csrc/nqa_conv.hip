// VGG conv3x3 + bias + ReLU for gfx950 (replaces the torchvision Conv2d+ReLU pairs the
// reference slices into stage1..5, nerf_qa/DISTS_pytorch/DISTS_pt.py:36-49,93-101).
//
// Two kernels:
//   conv1_1_kernel      3->64, fp32 VALU direct convolution with the (x-mean)/std input
//                       normalisation (DISTS_pt.py:92) in front.  K = 27 is too thin for
//                       MFMA and the input is fp32 NCHW, so this stays exact fp32.
//   conv3x3_igemm       layers 1..12: im2col-free implicit GEMM on MFMA.  Rows of the
//                       MFMA tile are output channels, columns are output pixels, the
//                       contraction runs over (tap, cin).  A block stages one halo tile of
//                       input pixels per 64-byte channel chunk in LDS and re-reads it for
//                       all 9 taps; weights arrive pre-tiled and pre-swizzled
//                       (nqa_pack_vgg_weights) one kernel row (3 taps) at a time.
//
// LDS images are rows of 64 bytes (one pixel or one output channel, 4 chunks of 16 B).
// Chunk c of row r sits at position c ^ ((r>>2)&3): the 16 lanes that ds_read_b128 serves
// together read 16 consecutive rows at one logical chunk, and the XOR spreads them over
// all 16 slots of the 256-byte bank row (cdna guide section 2 / T2).
#include "nqa_common.h"

namespace nqa {

// ---------------------------------------------------------------------------------
// conv1_1
// ---------------------------------------------------------------------------------
template <typename P>
__global__ __launch_bounds__(256) void conv1_1_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias,
                                                      typename P::T *__restrict__ out, int H, int W) {
  typedef typename P::T T;
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  const int py = pix / W, px = pix - py * W;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  float in[27];
  const float *xi = x + (size_t)n * 3 * HW;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int gy = py + ky - 1, gx = px + kx - 1;
      const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (ok) v = (xi[(size_t)c * HW + gy * W + gx] - mean[c]) / sd[c];
        in[(ky * 3 + kx) * 3 + c] = v;
      }
    }
  }
  T *o = out + ((size_t)n * HW + pix) * 64;
  // 4 groups of 16 output channels keeps the accumulators + inputs under 64 VGPRs.
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[g * 16 + j];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(in[k], w[k * 64 + g * 16 + j], acc[j]);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        f32x4 v = {fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f), fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f)};
        *reinterpret_cast<f32x4 *>(o + g * 16 + j) = v;
      }
    } else {
      typedef __attribute__((ext_vector_type(8))) T t8;
#pragma unroll
      for (int j = 0; j < 16; j += 8) {
        t8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = P::from_f(fmaxf(acc[j + e], 0.f));
        *reinterpret_cast<t8 *>(o + g * 16 + j) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// implicit-GEMM conv, layers 1..12
// ---------------------------------------------------------------------------------
template <int BN, int TW>
struct ConvGeom {
  static constexpr int MP = (BN == 128) ? 128 : 256;  // output pixels per block
  static constexpr int TH = MP / TW;                  // tile rows
  static constexpr int HW_ = TW + 2, HH_ = TH + 2;    // halo extent
  static constexpr int NQ = HW_ * HH_;                // halo pixels
  static constexpr int A_BYTES = NQ * 64;
  static constexpr int A_ITEMS = NQ * 4;
  static constexpr int A_ROUNDS = (A_ITEMS + 255) / 256;
  static constexpr int W_ROWS = 3 * BN;               // one kernel row of taps
  static constexpr int W_BYTES = W_ROWS * 64;
  static constexpr int W_ROUNDS = (W_ROWS * 4) / 256;
  static constexpr int LDS_BYTES = A_BYTES + W_BYTES;
};

template <typename P, int BN, int TW>
__global__ __launch_bounds__(256) void conv3x3_igemm_kernel(const typename P::T *__restrict__ in,
                                                            const char *__restrict__ wpk,
                                                            const float *__restrict__ bias,
                                                            typename P::T *__restrict__ out, int H, int W, int Cin,
                                                            int Cout, int tiles_x) {
  typedef typename P::T T;
  typedef ConvGeom<BN, TW> G;
  __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const int n = blockIdx.y, ct = blockIdx.z;
  const int x0 = bx * TW, y0 = by * G::TH;
  const int wn = (BN == 128) ? (wave & 1) : 0;   // which 64 output channels of the tile
  const int wm = (BN == 128) ? (wave >> 1) : wave;  // which 64 pixels of the tile
  const int nCC = Cin / P::KC;

  // ---- per-thread staging plan for the halo tile (constant across channel chunks) ----
  int a_goff[G::A_ROUNDS];
#pragma unroll
  for (int r = 0; r < G::A_ROUNDS; ++r) {
    const int i = tid + 256 * r;
    int off = -2;
    if (i < G::A_ITEMS) {
      const int q = i >> 2, c = (i & 3) ^ ((q >> 2) & 3);
      const int hy = q / G::HW_, hx = q - hy * G::HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      off = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (gy * W + gx) * Cin + c * P::CPC : -1;
    }
    a_goff[r] = off;
  }
  const T *in_img = in + (size_t)n * H * W * Cin;

  // ---- per-lane LDS read addresses ----
  int w_base[2], w_sw[2];  // weight rows (MFMA rows = output channels)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wn * 64 + i * 32 + l31;
    w_base[i] = G::A_BYTES + row * 64;
    w_sw[i] = (row >> 2) & 3;
  }
  int q0[2];  // halo index of this lane's output pixel at tap (0,0)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = (wm * 2 + j) * 32 + l31;
    const int ty = m / TW, tx = m - ty * TW;
    q0[j] = ty * G::HW_ + tx;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int cc = 0; cc < nCC; ++cc) {
    __syncthreads();  // everyone is done reading the previous halo tile and weight rows
    {
      const T *src = in_img + cc * P::KC;
#pragma unroll
      for (int r = 0; r < G::A_ROUNDS; ++r) {
        const int off = a_goff[r];
        if (off != -2) {
          u32x4 v = zero4;
          if (off >= 0) v = *reinterpret_cast<const u32x4 *>(src + off);
          *reinterpret_cast<u32x4 *>(smem + (tid + 256 * r) * 16) = v;
        }
      }
    }
    const u32x4 *slab = reinterpret_cast<const u32x4 *>(wpk + (size_t)(ct * nCC + cc) * 9 * BN * 64);
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      if (ky) __syncthreads();  // previous kernel row's weights fully consumed
#pragma unroll
      for (int r = 0; r < G::W_ROUNDS; ++r) {
        const int i = tid + 256 * r;
        reinterpret_cast<u32x4 *>(smem + G::A_BYTES)[i] = slab[ky * G::W_ROWS * 4 + i];
      }
      __syncthreads();
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int p_base[2], p_sw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = q0[j] + ky * G::HW_ + kx;
          p_base[j] = q * 64;
          p_sw[j] = (q >> 2) & 3;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ch = 2 * s + h;
          u32x4 af[2], bf[2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            af[i] = *reinterpret_cast<const u32x4 *>(smem + w_base[i] + kx * BN * 64 + ((ch ^ w_sw[i]) << 4));
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bf[j] = *reinterpret_cast<const u32x4 *>(smem + p_base[j] + ((ch ^ p_sw[j]) << 4));
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
        }
      }
    }
  }

  // ---- epilogue: bias + ReLU, 4 consecutive output channels per lane per store ----
  // acc[i][j][r]: channel = 8*(r>>2) + 4*h + (r&3) of row tile i, pixel = lane&31 of column tile j.
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = (wm * 2 + j) * 32 + l31;
    const int ty = m / TW, tx = m - ty * TW;
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < H && gx < W) {
      T *o = out + ((size_t)(n * H + gy) * W + gx) * Cout + ct * BN + wn * 64 + 4 * h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = i * 32 + 8 * g;
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + ct * BN + wn * 64 + 4 * h + co);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][4 * g + e] + b4[e], 0.f);
          if constexpr (sizeof(T) == 4) {
            f32x4 s4 = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(o + co) = s4;
          } else {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 s4 = {P::from_f(v[0]), P::from_f(v[1]), P::from_f(v[2]), P::from_f(v[3])};
            *reinterpret_cast<t4 *>(o + co) = s4;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------
template <typename P>
static int launch_conv1_1(const float *x, int n, int H, int W, const char *packed, void *out, hipStream_t st) {
  const float *w = reinterpret_cast<const float *>(packed + layer_offset(0, P::ID));
  const float *b = reinterpret_cast<const float *>(packed + layer_bias_offset(0, P::ID));
  dim3 grid(cdiv(H * W, 256), n);
  TimedLaunch t(NQA_K_CONV1, st);
  conv1_1_kernel<P><<<grid, 256, 0, st>>>(x, w, b, reinterpret_cast<typename P::T *>(out), H, W);
  return check_launch("conv1_1");
}

template <typename P, int BN, int TW>
static int launch_igemm(const void *in, int n, int H, int W, int cin, int cout, const char *wpk, const float *bias,
                        void *out, hipStream_t st) {
  typedef ConvGeom<BN, TW> G;
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, G::TH);
  dim3 grid(tiles_x * tiles_y, n, cout / BN);
  TimedLaunch t(NQA_K_CONV, st);
  conv3x3_igemm_kernel<P, BN, TW><<<grid, 256, 0, st>>>(reinterpret_cast<const typename P::T *>(in), wpk, bias,
                                                        reinterpret_cast<typename P::T *>(out), H, W, cin, cout,
                                                        tiles_x);
  return check_launch("conv3x3_igemm");
}

template <typename P>
static int launch_conv(const void *in, int n, int H, int W, int layer, const char *packed, void *out, hipStream_t st) {
  const ConvSpec &cs = kConvs[layer];
  const char *wpk = packed + layer_offset(layer, P::ID);
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, P::ID));
  const bool narrow = W <= 16;  // 32-wide tiles would be half empty
  if (conv_bn(cs.cout) == 128) {
    return narrow ? launch_igemm<P, 128, 16>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, st)
                  : launch_igemm<P, 128, 32>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, st);
  }
  return narrow ? launch_igemm<P, 64, 16>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, st)
                : launch_igemm<P, 64, 32>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, st);
}

int conv1_1(const float *x, int n, int H, int W, const void *packed, int prec, void *out, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  switch (prec) {
    case NQA_PREC_F32: return launch_conv1_1<PrecF32>(x, n, H, W, p, out, st);
    case NQA_PREC_BF16: return launch_conv1_1<PrecBF16>(x, n, H, W, p, out, st);
    case NQA_PREC_F16: return launch_conv1_1<PrecF16>(x, n, H, W, p, out, st);
  }
  set_error("conv1_1: unknown prec %d", prec);
  return NQA_E_ARG;
}

int conv3x3(const void *in, int n, int H, int W, int layer, const void *packed, int prec, void *out, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  switch (prec) {
    case NQA_PREC_F32: return launch_conv<PrecF32>(in, n, H, W, layer, p, out, st);
    case NQA_PREC_BF16: return launch_conv<PrecBF16>(in, n, H, W, layer, p, out, st);
    case NQA_PREC_F16: return launch_conv<PrecF16>(in, n, H, W, layer, p, out, st);
  }
  set_error("conv3x3: unknown prec %d", prec);
  return NQA_E_ARG;
}

}  // namespace nqa
