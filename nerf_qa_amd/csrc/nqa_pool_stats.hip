// HBM-bound kernels of the DISTS path for gfx950: L2-pool, per-channel statistics,
// similarity finalisation, alpha/beta score, and the NHWC -> NCHW export.
//
//   l2pool_kernel        L2pooling.forward, nerf_qa/DISTS_pytorch/DISTS_pt.py:22-25
//   stats_nhwc_kernel    the five sums behind DISTS_pt.py:131-139 on 16-byte channel groups
//   stats_nchw_kernel    same sums on float32 NCHW planes (stage 0 = raw image, and the
//                        forward_from_feats entry, DISTS_pt.py:181-202)
//   finalize_kernel      mean / variance / covariance -> S1, S2 (DISTS_pt.py:134,141)
//   score_kernel         alpha/beta weighted sum -> 1 - (dist1+dist2) (DISTS_pt.py:127-144)
//
// Statistics: per-thread shifted fp32 moments, fp64 from the block reduction on (see
// ShiftedMoments); the NCHW plane kernel (3-channel raw image, forward_from_feats) is fp64
// throughout.
#include "nqa_common.h"

namespace nqa {

// ---------------------------------------------------------------------------------
template <typename P>
__global__ __launch_bounds__(256) void l2pool_kernel(const typename P::T *__restrict__ in,
                                                     typename P::TO *__restrict__ out, int H, int W, int C, int Ho,
                                                     int Wo, long total) {
  typedef typename P::T T;
  typedef __attribute__((ext_vector_type(P::CPC))) T tvec;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int G = C / P::CPC;
  const int g = (int)(idx % G);
  long t = idx / G;
  const int ox = (int)(t % Wo);
  t /= Wo;
  const int oy = (int)(t % Ho);
  const int n = (int)(t / Ho);
  float acc[P::CPC];
#pragma unroll
  for (int e = 0; e < P::CPC; ++e) acc[e] = 0.f;
  const T *base = in + (size_t)n * H * W * C + g * P::CPC;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = 2 * oy - 1 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = 2 * ox - 1 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float wgt = ((dy == 1) ? 0.5f : 0.25f) * ((dx == 1) ? 0.5f : 0.25f);
      const tvec v = *reinterpret_cast<const tvec *>(base + ((size_t)iy * W + ix) * C);
#pragma unroll
      for (int e = 0; e < P::CPC; ++e) {
        const float f = P::to_f(v[e]);
        acc[e] = fmaf(f * f, wgt, acc[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < P::CPC; ++e) acc[e] = sqrtf(acc[e] + 1e-12f);
  store_pooled<P>(out + (((size_t)n * Ho + oy) * Wo + ox) * C, g * P::CPC, acc);
}

// ---------------------------------------------------------------------------------
// Statistics accumulation.  The reference takes the variance by a second pass over (f - mean)
// (DISTS_pt.py:137-138); a one-pass sum of f^2 in fp32 would cancel catastrophically for a
// channel whose spread is small next to its mean.  Each thread therefore accumulates SHIFTED
// moments in fp32 -- sum(f-p), sum((f-p)^2), sum((fx-px)(fy-py)) with the pivot p = the
// thread's first sample of that channel, so the sums are of the order of the variance -- and
// converts them to raw fp64 sums once, at the end; block partials and the final combine are
// fp64.  Every feature byte is read once and the error stays relative to the variance.
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int N>
struct ShiftedMoments {  // N channels held as N/2 float pairs, so every update is packed (v_pk_*_f32)
  f32x2 px[N / 2], py[N / 2];  // pivots
  f32x2 s1x[N / 2], s1y[N / 2], s2x[N / 2], s2y[N / 2], sxy[N / 2];
  int n;
  __device__ inline void init() {
    n = 0;
#pragma unroll
    for (int e = 0; e < N / 2; ++e) px[e] = py[e] = s1x[e] = s1y[e] = s2x[e] = s2y[e] = sxy[e] = (f32x2){0.f, 0.f};
  }
  __device__ inline void add2(int e2, f32x2 x, f32x2 y) {
    const f32x2 dx = x - px[e2], dy = y - py[e2];
    s1x[e2] += dx;
    s1y[e2] += dy;
    s2x[e2] = dx * dx + s2x[e2];
    s2y[e2] = dy * dy + s2y[e2];
    sxy[e2] = dx * dy + sxy[e2];
  }
  // raw sum s of channel e, s = {sum x, sum y, sum x^2, sum y^2, sum xy}
  __device__ inline double raw(int e, int s) const {
    const int e2 = e >> 1, k = e & 1;
    const double p = px[e2][k], q = py[e2][k], nn = n, ax = s1x[e2][k], ay = s1y[e2][k];
    switch (s) {
      case 0: return ax + nn * p;
      case 1: return ay + nn * q;
      case 2: return (double)s2x[e2][k] + 2.0 * p * ax + nn * p * p;
      case 3: return (double)s2y[e2][k] + 2.0 * q * ay + nn * q * q;
      default: return (double)sxy[e2][k] + q * ax + p * ay + nn * p * q;
    }
  }
};
// one loaded 16-byte channel group as float pairs
template <typename P, typename V>
__device__ inline void unpack2(const V &v, f32x2 (&out)[P::CPC / 2]) {
#pragma unroll
  for (int e = 0; e < P::CPC / 2; ++e) out[e] = (f32x2){P::to_f(v[2 * e]), P::to_f(v[2 * e + 1])};
}

// Block reduction of the per-thread raw sums over the pixel lanes -> part[(b*nblk+blk)*C*5 ...].
template <int CPC>
__device__ inline void reduce_store(const ShiftedMoments<CPC> &m, double *red, int tid, int G, int PL, int C,
                                    double *dst) {
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CPC; ++e) red[tid * CPC + e] = m.raw(e, s);
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      const int cg = c / CPC, ce = c % CPC;
      double sum = 0.0;
      for (int q = 0; q < PL; ++q) sum += red[(q * G + cg) * CPC + ce];
      dst[(size_t)c * 5 + s] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------
// L2-pool and the five statistics sums of one tapped map in ONE pass over it.  A block owns a
// strip of OUTPUT pixels of image pair b (x image b, y image B+b); a thread owns one 16-byte
// channel group and walks the strip.  For each output pixel it loads the 3x3 window of x and
// of y, writes both pooled pixels, and adds the window's lower-right 2x2 (input pixels
// (2oy..2oy+1, 2ox..2ox+1), each owned by exactly one output pixel) to the sums.
// (three blocks per CU for the float instances, 128 VGPRs; the 16-bit instances need 176 with the row-reuse
// registers and would spill 12 B/lane at three -- they run two blocks per CU, which measured the same at 1080p)
#ifdef NQA_POOL_NT  // A/B build: the tap is read once by this pass -> non-temporal loads
#define NQA_TAP_LOAD(p) __builtin_nontemporal_load(p)
#else
#define NQA_TAP_LOAD(p) (*(p))
#endif
template <typename P>
__global__ __launch_bounds__(256, sizeof(typename P::T) == 2 ? 2 : 3) void pool_stats_kernel(const typename P::T *__restrict__ feat,
                                                         typename P::TO *__restrict__ pooled, int B, int H, int W,
                                                         int C, int Ho, int Wo, int TR, int TC, int tiles_x,
                                                         int nblk, double *__restrict__ part) {
  typedef typename P::T T;
  typedef __attribute__((ext_vector_type(P::CPC))) T tvec;
  __shared__ double red[256 * P::CPC];
  const int tid = threadIdx.x;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so ids equal mod 8 share an L2;
  // each class gets a contiguous run of tiles, and the input row / column that two neighbouring tiles' 3x3
  // windows share is fetched from HBM once instead of once per L2 (speed only; any placement is correct)
  int id = blockIdx.x;
  {
    const int nb = gridDim.x, qq = nb >> 3, rr = nb & 7, xcd = id & 7, local = id >> 3;
    id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
  }
  const int b = id / nblk, blk = id - b * nblk;
  const int G = C / P::CPC, PL = 256 / G;
  const int g = tid % G, pl = tid / G;
  const int HoWo = Ho * Wo;
  // the block's output pixels form a TR x TC tile (not a 1-D strip): the 3x3 windows of vertically
  // adjacent outputs share an input row, and inside one block that re-read hits L1/L2 instead of HBM
  const int by = blk / tiles_x, bx = blk - by * tiles_x;
  const int oy0 = by * TR, ox0 = bx * TC, tile_units = TR * TC;
  const T *fx = feat + (size_t)b * H * W * C + g * P::CPC;
  const T *fy = feat + (size_t)(B + b) * H * W * C + g * P::CPC;
  typename P::TO *ox_ = pooled + (size_t)b * HoWo * C;
  typename P::TO *oy_ = pooled + (size_t)(B + b) * HoWo * C;
  ShiftedMoments<P::CPC> m;
  m.init();
  if (pl < tile_units) {  // pivot = a sample near this thread's first pixel (a window centre, always in range)
    const int oy = min(oy0 + pl / TC, Ho - 1), ox = min(ox0 + pl % TC, Wo - 1);
    const size_t o = ((size_t)(2 * oy) * W + 2 * ox) * C;
    const tvec vx = *reinterpret_cast<const tvec *>(fx + o), vy = *reinterpret_cast<const tvec *>(fy + o);
    unpack2<P>(vx, m.px);
    unpack2<P>(vy, m.py);
  }
  tvec keepx[3], keepy[3];  // bottom row of the previous window of this thread
  bool have_prev = false;
  for (int t = pl; t < tile_units; t += PL) {
    const int oy = oy0 + t / TC, ox = ox0 + t % TC;
    if (oy >= Ho || ox >= Wo) {
      have_prev = false;
      continue;
    }
    const int u = oy * Wo + ox;
    // All 18 loads go out unconditionally, back to back (a branch per tap would serialise them into
    // 9 round trips).  A window that lies wholly inside the image -- almost all of them -- takes the
    // fast path: addresses are one base plus constants, every tap is live, no selects; at the image
    // border the coordinates are clamped, and an out-of-image tap is zeroed for the pool (zero
    // padding) and replaced by the pivot for the sums (it then adds exactly 0).
    const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
    const bool interior = iy0 >= 0 && iy0 + 2 < H && ix0 >= 0 && ix0 + 2 < W;
    tvec vx[9], vy[9];
    if (interior) {
      const size_t o0 = ((size_t)iy0 * W + ix0) * C;
      // the window's top row is the previous window's bottom row when this thread's previous pooled pixel was the
      // one directly above (tiles are one pass wide, so that is the previous iteration) and was interior too
      const bool reuse = TC == PL && have_prev;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const size_t o = o0 + ((size_t)(t / 3) * W + (t % 3)) * C;
        if (t < 3 && reuse) {
          vx[t] = keepx[t];
          vy[t] = keepy[t];
        } else {
          vx[t] = NQA_TAP_LOAD(reinterpret_cast<const tvec *>(fx + o));
          vy[t] = NQA_TAP_LOAD(reinterpret_cast<const tvec *>(fy + o));
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int cy = min(max(iy0 + t / 3, 0), H - 1), cx = min(max(ix0 + t % 3, 0), W - 1);
        const size_t o = ((size_t)cy * W + cx) * C;
        vx[t] = *reinterpret_cast<const tvec *>(fx + o);
        vy[t] = *reinterpret_cast<const tvec *>(fy + o);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the 18 loads ahead of all the arithmetic
    f32x2 qx[P::CPC / 2], qy[P::CPC / 2];
#pragma unroll
    for (int e = 0; e < P::CPC / 2; ++e) qx[e] = qy[e] = (f32x2){0.f, 0.f};
    if (interior) {
      m.n += 4;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const float wgt = ((dy == 1) ? 0.5f : 0.25f) * ((dx == 1) ? 0.5f : 0.25f);
        f32x2 X[P::CPC / 2], Y[P::CPC / 2];
        unpack2<P>(vx[t], X);
        unpack2<P>(vy[t], Y);
#pragma unroll
        for (int e = 0; e < P::CPC / 2; ++e) {
          qx[e] = (X[e] * X[e]) * wgt + qx[e];
          qy[e] = (Y[e] * Y[e]) * wgt + qy[e];
          if (dy >= 1 && dx >= 1) m.add2(e, X[e], Y[e]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const bool ok = (unsigned)(iy0 + dy) < (unsigned)H && (unsigned)(ix0 + dx) < (unsigned)W;
        const float wgt = ok ? ((dy == 1) ? 0.5f : 0.25f) * ((dx == 1) ? 0.5f : 0.25f) : 0.f;
        if (dy >= 1 && dx >= 1 && ok) m.n += 1;
        f32x2 X[P::CPC / 2], Y[P::CPC / 2];
        unpack2<P>(vx[t], X);
        unpack2<P>(vy[t], Y);
#pragma unroll
        for (int e = 0; e < P::CPC / 2; ++e) {
          qx[e] = (X[e] * X[e]) * wgt + qx[e];
          qy[e] = (Y[e] * Y[e]) * wgt + qy[e];
          if (dy >= 1 && dx >= 1) m.add2(e, ok ? X[e] : m.px[e], ok ? Y[e] : m.py[e]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      keepx[t] = vx[6 + t];
      keepy[t] = vy[6 + t];
    }
    have_prev = interior;
    float px[P::CPC], py[P::CPC];
#pragma unroll
    for (int e = 0; e < P::CPC; ++e) {
      px[e] = sqrtf(qx[e >> 1][e & 1] + 1e-12f);
      py[e] = sqrtf(qy[e >> 1][e & 1] + 1e-12f);
    }
    store_pooled<P>(ox_ + (size_t)u * C, g * P::CPC, px);
    store_pooled<P>(oy_ + (size_t)u * C, g * P::CPC, py);
  }
  reduce_store<P::CPC>(m, red, tid, G, PL, C, part + ((size_t)b * nblk + blk) * C * 5);
}

// ---------------------------------------------------------------------------------
// Partial sums layout: part[((b*nblk + blk)*C + c)*5 + s], s = {sum x, sum y, sum x^2, sum y^2, sum xy}.
template <typename P>
__global__ __launch_bounds__(256) void stats_nhwc_kernel(const typename P::T *__restrict__ feat, int B, int HW, int C,
                                                         int pix_per_block, double *__restrict__ part) {
  typedef typename P::T T;
  typedef __attribute__((ext_vector_type(P::CPC))) T tvec;
  __shared__ double red[256 * P::CPC];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int G = C / P::CPC;   // 16-byte channel groups per pixel (<= 128)
  const int PL = 256 / G;     // pixels handled side by side
  const int g = tid % G, pl = tid / G;
  const int p_begin = blk * pix_per_block;
  const int p_end = min(HW, p_begin + pix_per_block);
  const T *fx = feat + (size_t)b * HW * C + g * P::CPC;
  const T *fy = feat + (size_t)(B + b) * HW * C + g * P::CPC;
  ShiftedMoments<P::CPC> m;
  m.init();
  if (p_begin + pl < p_end) {
    const tvec vx = *reinterpret_cast<const tvec *>(fx + (size_t)(p_begin + pl) * C);
    const tvec vy = *reinterpret_cast<const tvec *>(fy + (size_t)(p_begin + pl) * C);
    unpack2<P>(vx, m.px);
    unpack2<P>(vy, m.py);
  }
#pragma unroll 4
  for (int p = p_begin + pl; p < p_end; p += PL) {
    const tvec vx = *reinterpret_cast<const tvec *>(fx + (size_t)p * C);
    const tvec vy = *reinterpret_cast<const tvec *>(fy + (size_t)p * C);
    m.n += 1;
    f32x2 X[P::CPC / 2], Y[P::CPC / 2];
    unpack2<P>(vx, X);
    unpack2<P>(vy, Y);
#pragma unroll
    for (int e = 0; e < P::CPC / 2; ++e) m.add2(e, X[e], Y[e]);
  }
  reduce_store<P::CPC>(m, red, tid, G, PL, C, part + ((size_t)b * nblk + blk) * C * 5);
}

// float32 NCHW planes: grid (B*C, nblk).  fx, fy: (B, C, HW).
__global__ __launch_bounds__(256) void stats_nchw_kernel(const float *__restrict__ fx, const float *__restrict__ fy,
                                                         int C, int HW, int pix_per_block, double *__restrict__ part) {
  __shared__ double red[5][256];
  const int tid = threadIdx.x;
  const int bc = blockIdx.x, blk = blockIdx.y, nblk = gridDim.y;
  const int b = bc / C, c = bc - b * C;
  const float *px = fx + (size_t)bc * HW, *py = fy + (size_t)bc * HW;
  const int p_begin = blk * pix_per_block;
  const int p_end = min(HW, p_begin + pix_per_block);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll 4
  for (int p = p_begin + tid; p < p_end; p += 256) {
    const double x = (double)px[p], y = (double)py[p];
    a0 += x;
    a1 += y;
    a2 = fma(x, x, a2);
    a3 = fma(y, y, a3);
    a4 = fma(x, y, a4);
  }
  red[0][tid] = a0;
  red[1][tid] = a1;
  red[2][tid] = a2;
  red[3][tid] = a3;
  red[4][tid] = a4;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int s = 0; s < 5; ++s) red[s][tid] += red[s][tid + off];
    }
    __syncthreads();
  }
  if (tid < 5) part[(((size_t)b * nblk + blk) * C + c) * 5 + tid] = red[tid][0];
}

// ---------------------------------------------------------------------------------
// One wave per (pair, channel): lanes stride over the per-block partial sums, then a shuffle
// tree; the mean / variance / covariance -> S1, S2 arithmetic is done in fp64.
__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ part, StageDesc d,
                                                       float *__restrict__ s1, float *__restrict__ s2) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int gc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gc >= d.ctot) return;
  int k = 0;
  while (k + 1 < d.nstage && gc >= d.coff[k + 1]) ++k;
  const int c = gc - d.coff[k];
  const double *p = part + d.part_off[k] + ((size_t)b * d.nblk[k] * d.c[k] + c) * 5;
  double s[5] = {0, 0, 0, 0, 0};
  for (int blk = lane; blk < d.nblk[k]; blk += 64) {
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] += p[(size_t)blk * d.c[k] * 5 + q];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q)
    for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
  if (lane) return;
  const double inv = 1.0 / (double)d.hw[k];
  const double mx = s[0] * inv, my = s[1] * inv;
  const double vx = s[2] * inv - mx * mx, vy = s[3] * inv - my * my;
  const double cov = s[4] * inv - mx * my;
  const double c1 = 1e-6, c2 = 1e-6;
  s1[(size_t)b * d.ctot + gc] = (float)((2.0 * mx * my + c1) / (mx * mx + my * my + c1));
  s2[(size_t)b * d.ctot + gc] = (float)((2.0 * cov + c2) / (vx + vy + c2));
}

// one block per pair: score_b = 1 - sum_c (alpha_c*S1 + beta_c*S2)/w
__global__ __launch_bounds__(256) void score_kernel(const float *__restrict__ s1, const float *__restrict__ s2,
                                                    const float *__restrict__ alpha, const float *__restrict__ beta,
                                                    int ctot, float *__restrict__ score) {
  __shared__ double red[2][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double w = 0.0, d = 0.0;
  for (int c = tid; c < ctot; c += 256) {
    const double a = alpha[c], be = beta[c];
    w += a + be;
    d += a * (double)s1[(size_t)b * ctot + c] + be * (double)s2[(size_t)b * ctot + c];
  }
  red[0][tid] = w;
  red[1][tid] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      red[0][tid] += red[0][tid + off];
      red[1][tid] += red[1][tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) score[b] = (float)(1.0 - red[1][0] / red[0][0]);
}

// ---------------------------------------------------------------------------------
// NHWC (T) -> NCHW float32 through a 64-pixel x 64-channel LDS tile.
template <typename P>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const typename P::T *__restrict__ in,
                                                           float *__restrict__ out, int HW, int C) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? P::to_f(in[((size_t)n * HW + p) * C + c]) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) out[((size_t)n * C + c) * HW + p] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------
template <typename P>
static int launch_l2pool(const void *in, int n, int H, int W, int C, void *out, hipStream_t st) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)n * Ho * Wo * (C / P::CPC);
  TimedLaunch t(NQA_K_POOL, st);
  l2pool_kernel<P><<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(
      reinterpret_cast<const typename P::T *>(in), reinterpret_cast<typename P::TO *>(out), H, W, C, Ho, Wo, total);
  return check_launch("l2pool");
}
int l2pool_to_split16(const void *in_f16, int n, int H, int W, int C, void *out_split16, hipStream_t st) {
  return launch_l2pool<PrecF16X>(in_f16, n, H, W, C, out_split16, st);
}

int l2pool(const void *in, int n, int H, int W, int C, int prec, void *out, hipStream_t st) {
  switch (prec) {
    case NQA_PREC_F32S: return launch_l2pool<PrecF32S>(in, n, H, W, C, out, st);  // float in, split16 out
    case NQA_PREC_F32: return launch_l2pool<PrecF32>(in, n, H, W, C, out, st);
    case NQA_PREC_BF16: return launch_l2pool<PrecBF16>(in, n, H, W, C, out, st);
    case NQA_PREC_F16: return launch_l2pool<PrecF16>(in, n, H, W, C, out, st);
  }
  set_error("l2pool: unknown prec %d", prec);
  return NQA_E_ARG;
}

// Work split of the NHWC statistics kernels: `units` loop items (pixels, or output pixels for
// the fused pool+stats pass) per image pair; each block takes a contiguous strip.  Threads
// take 4..16 items each (every block also writes C*5 doubles of partial sums, so very short
// strips would cost more in partials than they read), sized for ~1024 blocks over the batch,
// and never more than ~4096 so the finalize pass stays short.
int stats_units_per_block(int units, int C, int prec, int B) {
  prec = storage_prec(prec);
  const int cpc = prec == NQA_PREC_F32 ? 4 : 8;
  const int PL = 256 / (C / cpc);
  if (B < 1) B = 1;
  // ~1024 blocks over the batch, fewer for wide taps: every block leaves C*5 doubles of partial sums and
  // the finalize pass reads them all (at C = 512, 1024 blocks are 21 MB of partials per tap)
  const long target_blocks = C >= 512 ? 384 : (C >= 256 ? 768 : 1024);
  long per_thread = (long)units * B / ((long)PL * target_blocks);
  per_thread = per_thread < 4 ? 4 : (per_thread > 16 ? 16 : per_thread);
  int upb = (int)per_thread * PL;
  const int max_blocks = 4096 / B > 16 ? 4096 / B : 16;
  if (cdiv(units, upb) > max_blocks) upb = cdiv(cdiv(units, max_blocks), PL) * PL;
  return upb;
}
// 4096 pixels (16 per thread) per block: a 256x256 plane is 16 blocks, so even B = 1 (3 planes per tensor)
// puts 48 blocks on the chip instead of 3 threads-serial ones (25 -> 6 us), and B = 32 fills it
int stats_nchw_ppb(int HW) { return HW > 4096 ? 4096 : (HW > 0 ? HW : 1); }

template <typename P>
static int launch_stats_nhwc(const void *feat, int B, int HW, int C, double *part, hipStream_t st) {
  const int ppb = stats_units_per_block(HW, C, P::ID, B);
  dim3 grid(cdiv(HW, ppb), B);
  TimedLaunch t(NQA_K_STATS, st);
  stats_nhwc_kernel<P><<<grid, 256, 0, st>>>(reinterpret_cast<const typename P::T *>(feat), B, HW, C, ppb, part);
  return check_launch("stats_nhwc");
}

int stats_nhwc(const void *feat, int B, int HW, int C, int prec, double *part, hipStream_t st) {
  prec = storage_prec(prec);
  switch (prec) {
    case NQA_PREC_F32: return launch_stats_nhwc<PrecF32>(feat, B, HW, C, part, st);
    case NQA_PREC_BF16: return launch_stats_nhwc<PrecBF16>(feat, B, HW, C, part, st);
    case NQA_PREC_F16: return launch_stats_nhwc<PrecF16>(feat, B, HW, C, part, st);
  }
  set_error("stats: unknown prec %d", prec);
  return NQA_E_ARG;
}

// Tile shape of the fused pool+statistics pass for an Ho x Wo pooled map: units_per_block pooled
// pixels as a NARROW, TALL tile.  Every odd input row serves two pooled rows; a block walks its
// tile row by row, so with a narrow tile the second use follows the first within a pass or two and
// hits L1/L2.  (With wide 8-row tiles the row had left the XCD's 4 MB L2 by then: rocprof FETCH_SIZE
// read 1.5x the algorithmic bytes on every tap.)  The width keeps one input row of the tile near
// 8 KB and is a multiple of the pixels the block covers per pass.
int pool_stats_tiles(int Ho, int Wo, int C, int prec, int B, int *tr, int *tc) {
  prec = storage_prec(prec);
  const int upb = stats_units_per_block(Ho * Wo, C, prec, B);
  const int esz = (int)prec_elem_bytes(prec), cpc = 16 / esz;
  const int PL = 256 / (C / cpc);  // pooled pixels per pass of the block
  // exactly one pass wide: a thread then walks ONE column of the tile from top to bottom and can keep the input
  // row that two vertically adjacent 3x3 windows share in registers (6 + 6 loads per pooled pixel instead of 9 + 9)
  int TC = PL;
  int TR = upb / TC;
  if (TR < 1) TR = 1;
  if (TR > Ho) TR = Ho;
  if (tr) *tr = TR;
  if (tc) *tc = TC;
  return cdiv(Wo, TC) * cdiv(Ho, TR);
}

template <typename P>
static int launch_pool_stats(const void *feat, int B, int H, int W, int C, void *pooled, double *part,
                             hipStream_t st) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int TR, TC;
  const int nblk = pool_stats_tiles(Ho, Wo, C, P::ID, B, &TR, &TC);
  TimedLaunch t(NQA_K_POOL, st);
  pool_stats_kernel<P><<<nblk * B, 256, 0, st>>>(reinterpret_cast<const typename P::T *>(feat),
                                                 reinterpret_cast<typename P::TO *>(pooled), B, H, W, C, Ho, Wo, TR, TC,
                                                 cdiv(Wo, TC), nblk, part);
  return check_launch("pool_stats");
}
int pool_stats_to_split16(const void *feat_f16, int B, int H, int W, int C, void *pooled_split16, double *part,
                          hipStream_t st) {
  return launch_pool_stats<PrecF16X>(feat_f16, B, H, W, C, pooled_split16, part, st);
}

// tap (2B images: x then y) -> pooled (2B images) + statistics partials of the B pairs
int pool_stats(const void *feat, int B, int H, int W, int C, int prec, void *pooled, double *part, hipStream_t st) {
  switch (prec) {
    case NQA_PREC_F32S: return launch_pool_stats<PrecF32S>(feat, B, H, W, C, pooled, part, st);  // split16 out
    case NQA_PREC_F32: return launch_pool_stats<PrecF32>(feat, B, H, W, C, pooled, part, st);
    case NQA_PREC_BF16: return launch_pool_stats<PrecBF16>(feat, B, H, W, C, pooled, part, st);
    case NQA_PREC_F16: return launch_pool_stats<PrecF16>(feat, B, H, W, C, pooled, part, st);
  }
  set_error("pool_stats: unknown prec %d", prec);
  return NQA_E_ARG;
}

int stats_nchw(const float *fx, const float *fy, int B, int C, int HW, double *part, hipStream_t st) {
  const int ppb = stats_nchw_ppb(HW);
  dim3 grid(B * C, cdiv(HW, ppb));
  TimedLaunch t(NQA_K_STATS, st);
  stats_nchw_kernel<<<grid, 256, 0, st>>>(fx, fy, C, HW, ppb, part);
  return check_launch("stats_nchw");
}

int finalize(const double *part, const StageDesc &d, int B, float *s1, float *s2, hipStream_t st) {
  dim3 grid(cdiv(d.ctot, 4), B);
  TimedLaunch t(NQA_K_STATS, st);
  finalize_kernel<<<grid, 256, 0, st>>>(part, d, s1, s2);
  return check_launch("finalize");
}

int score(const float *s1, const float *s2, const float *alpha, const float *beta, int B, float *out,
          hipStream_t st) {
  score_kernel<<<B, 256, 0, st>>>(s1, s2, alpha, beta, NQA_TOTAL_CHNS, out);
  return check_launch("score");
}

template <typename P>
static int launch_export(const void *in, int n, int HW, int C, float *out, hipStream_t st) {
  dim3 grid(cdiv(HW, 64), cdiv(C, 64), n);
  nhwc_to_nchw_kernel<P><<<grid, 256, 0, st>>>(reinterpret_cast<const typename P::T *>(in), out, HW, C);
  return check_launch("nhwc_to_nchw");
}

int nhwc_to_nchw(const void *in, int n, int HW, int C, int prec, float *out, hipStream_t st) {
  prec = storage_prec(prec);
  switch (prec) {
    case NQA_PREC_F32: return launch_export<PrecF32>(in, n, HW, C, out, st);
    case NQA_PREC_BF16: return launch_export<PrecBF16>(in, n, HW, C, out, st);
    case NQA_PREC_F16: return launch_export<PrecF16>(in, n, HW, C, out, st);
  }
  set_error("nhwc_to_nchw: unknown prec %d", prec);
  return NQA_E_ARG;
}

}  // namespace nqa
