// A-DISTS on gfx950: everything after the two VGG pyramids of ADISTS.forward
// (nerf_qa/ADISTS/ADISTS.py:137-197, as_map=False).
//
// The reference runs, per stage, 2 depthwise 21x21 Gaussian convolutions for the texture
// probability (compute_prob, :71-100) and 5 more for the windowed T/S maps (:165-183), on
// L2-normalised features.  Three observations shape the kernels here:
//   * the window is an outer product (:106-110), so each filter is two 21-tap passes;
//   * F.normalize(f, dim=(2,3)) is one scalar per (image, channel), so the windowed moments
//     of the normalised maps are the moments of the raw maps times inv_x, inv_y, inv_x^2 ...;
//     the raw x moments also give compute_prob's mean/variance -- one windowed pass serves both;
//   * D = sum_c w_c mean_hw((1-ps) T_c + ps S_c) = mean_hw((1-ps) TW + ps SW) with
//     TW = sum_c w_c T_c, SW = sum_c w_c S_c, so the heavy pass reduces over channels and
//     leaves three small (B,h-20,w-20) maps; the coarse-to-fine probability chain then
//     works on those maps only.
// Global per-(image,channel) sums come from the DISTS statistics kernels (fp64).
#include <math.h>
#include <string.h>

#include "nqa_common.h"

namespace nqa {

static constexpr int kWin = 21;
static constexpr int kChainBlocks = 1024;  // most blocks per image of the chain's reduction kernels

// tuning hook (nqa_set_conv_variant bit 3): the per-wave-loads form of the window pass, kept for A/B timing
static thread_local bool g_window_legacy = false;
void set_adists_window_legacy(bool on) { g_window_legacy = on; }
static bool adists_window_legacy() { return g_window_legacy; }

struct Gauss {
  float g[kWin];
};

// ---------------------------------------------------------------------------------
// (b,3,H,W) float32 NCHW -> (b,H,W,4) float32 NHWC with a zero 4th channel (stage 0)
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                             int HW) {
  const int p = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (p >= HW) return;
  const float *s = in + (size_t)n * 3 * HW;
  f32x4 v = {s[p], s[HW + p], s[2 * HW + p], 0.f};
  *reinterpret_cast<f32x4 *>(out + ((size_t)n * HW + p) * 4) = v;
}

// ---------------------------------------------------------------------------------
// Per (b, channel) scalars from the fp64 partial sums (layout of stats_*_kernel):
//   q[0]=inv_x  q[1]=inv_y   1/max(||f||_2, 1e-12)               (ADISTS.py:130,166-167)
//   q[2]=sum_x               for the entropy normalisation       (:132)
//   q[3..7]= mean_x mean_y var_x var_y cov of the RAW maps (population), for the global branch
__global__ __launch_bounds__(256) void adists_prep_kernel(const double *__restrict__ part, StageDesc d,
                                                          float *__restrict__ q /* [8][B][ctot] */, int B) {
  const int b = blockIdx.y;
  const int gc = blockIdx.x * 256 + threadIdx.x;
  if (gc >= d.ctot) return;
  int k = 0;
  while (k + 1 < d.nstage && gc >= d.coff[k + 1]) ++k;
  const int c = gc - d.coff[k];
  const double *p = part + d.part_off[k] + ((size_t)b * d.nblk[k] * d.c[k] + c) * 5;
  double s[5] = {0, 0, 0, 0, 0};
  for (int blk = 0; blk < d.nblk[k]; ++blk)
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] += p[(size_t)blk * d.c[k] * 5 + j];
  const double inv = 1.0 / (double)d.hw[k];
  const double mx = s[0] * inv, my = s[1] * inv;
  const size_t o = (size_t)b * d.ctot + gc, st = (size_t)B * d.ctot;
  q[0 * st + o] = (float)(1.0 / fmax(sqrt(s[2]), 1e-12));
  q[1 * st + o] = (float)(1.0 / fmax(sqrt(s[3]), 1e-12));
  q[2 * st + o] = (float)s[0];
  q[3 * st + o] = (float)mx;
  q[4 * st + o] = (float)my;
  q[5 * st + o] = (float)(s[2] * inv - mx * mx);
  q[6 * st + o] = (float)(s[3] * inv - my * my);
  q[7 * st + o] = (float)(s[4] * inv - mx * my);
}

// ---------------------------------------------------------------------------------
// Spatial entropy of the x maps (ADISTS.py:127-133): p = relu(f)*inv; p /= sum(p)+c0;
// H = -sum p log2(p + c0).  Partial sums per block, part[(b*nblk+blk)*C + c].
template <typename P>
__global__ __launch_bounds__(256) void entropy_nhwc_kernel(const typename P::T *__restrict__ feat, int HW, int C,
                                                           int pix_per_block, const float *__restrict__ invx,
                                                           const float *__restrict__ sumx, int ctot,
                                                           double *__restrict__ part) {
  typedef typename P::T T;
  typedef __attribute__((ext_vector_type(P::CPC))) T tvec;
  __shared__ double red[256 * P::CPC];
  const int tid = threadIdx.x, b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int G = C / P::CPC, PL = 256 / G;
  const int g = tid % G, pl = tid / G;
  float inv[P::CPC], den[P::CPC];
#pragma unroll
  for (int e = 0; e < P::CPC; ++e) {
    inv[e] = invx[(size_t)b * ctot + g * P::CPC + e];
    den[e] = inv[e] * sumx[(size_t)b * ctot + g * P::CPC + e] + 1e-12f;
  }
  double acc[P::CPC];
#pragma unroll
  for (int e = 0; e < P::CPC; ++e) acc[e] = 0.0;
  const T *f = feat + (size_t)b * HW * C + g * P::CPC;
  const int p_end = min(HW, (blk + 1) * pix_per_block);
  for (int p = blk * pix_per_block + pl; p < p_end; p += PL) {
    const tvec v = *reinterpret_cast<const tvec *>(f + (size_t)p * C);
#pragma unroll
    for (int e = 0; e < P::CPC; ++e) {
      const float pn = fmaxf(P::to_f(v[e]), 0.f) * inv[e] / den[e];
      acc[e] += (double)(-pn * log2f(pn + 1e-12f));
    }
  }
#pragma unroll
  for (int e = 0; e < P::CPC; ++e) red[tid * P::CPC + e] = acc[e];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int cg = c / P::CPC, ce = c % P::CPC;
    double s = 0.0;
    for (int q = 0; q < PL; ++q) s += red[(q * G + cg) * P::CPC + ce];
    part[((size_t)b * nblk + blk) * C + c] = s;
  }
}

// Channel weights (ADISTS.py:134-135,154-160): per stage H/(sum_stage H + c0)*C, then over
// all 1475: /sum, clamp to mean +- 0.5*population std, /sum.  One block per image.
struct EntDesc {
  long part_off[NQA_NUM_TAPS];
  int nblk[NQA_NUM_TAPS];
  int c[NQA_NUM_TAPS];      // padded channel count used by the partial layout (stage 0: 4)
  int creal[NQA_NUM_TAPS];  // real channels
  int coff[NQA_NUM_TAPS];
  int ctot;
};

// fold the per-block entropy partials: one wave per (image, channel) -> hsum[b][ctot]
__global__ __launch_bounds__(256) void adists_entropy_fold_kernel(const double *__restrict__ part, EntDesc d,
                                                                  float *__restrict__ hsum) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int gc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gc >= d.ctot) return;
  int k = 0;
  while (k + 1 < NQA_NUM_TAPS && gc >= d.coff[k + 1]) ++k;
  const int c = gc - d.coff[k];
  const double *p = part + d.part_off[k] + (size_t)b * d.nblk[k] * d.c[k] + c;
  double s = 0.0;
  for (int blk = lane; blk < d.nblk[k]; blk += 64) s += p[(size_t)blk * d.c[k]];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) hsum[(size_t)b * d.ctot + gc] = (float)s;
}

__global__ __launch_bounds__(256) void adists_weights_kernel(const float *__restrict__ hsum, EntDesc d,
                                                             float *__restrict__ wgt /* [B][ctot] */) {
  __shared__ float h[NQA_TOTAL_CHNS];
  __shared__ double red[256];
  __shared__ double bc[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  auto block_sum = [&](double v) -> double {
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    return red[0];
  };
  for (int k = 0; k < NQA_NUM_TAPS; ++k) {
    double loc = 0.0;
    for (int c = tid; c < d.creal[k]; c += 256) {
      const float s = hsum[(size_t)b * d.ctot + d.coff[k] + c];
      h[d.coff[k] + c] = s;
      loc += (double)s;
    }
    const float tot = (float)block_sum(loc);
    for (int c = tid; c < d.creal[k]; c += 256)
      h[d.coff[k] + c] = h[d.coff[k] + c] / (tot + 1e-12f) * (float)d.creal[k];
  }
  __syncthreads();
  double loc = 0.0;
  for (int c = tid; c < d.ctot; c += 256) loc += (double)h[c];
  const float s0 = (float)block_sum(loc);
  for (int c = tid; c < d.ctot; c += 256) h[c] = h[c] / s0;
  __syncthreads();
  loc = 0.0;
  for (int c = tid; c < d.ctot; c += 256) loc += (double)h[c];
  const float mean = (float)(block_sum(loc) / d.ctot);
  loc = 0.0;
  for (int c = tid; c < d.ctot; c += 256) {
    const float dv = h[c] - mean;
    loc += (double)(dv * dv);
  }
  const float sd = sqrtf((float)(block_sum(loc) / d.ctot));
  const float lo = mean - 0.5f * sd, hi = mean + 0.5f * sd;
  loc = 0.0;
  for (int c = tid; c < d.ctot; c += 256) {
    h[c] = fminf(fmaxf(h[c], lo), hi);
    loc += (double)h[c];
  }
  const float s1 = (float)block_sum(loc);
  (void)bc;
  for (int c = tid; c < d.ctot; c += 256) wgt[(size_t)b * d.ctot + c] = h[c] / s1;
}

// ---------------------------------------------------------------------------------
// The windowed pass for the NHWC taps (stages 1..5), lanes = channels.
//
// A wave owns one output column x of image pair b and a strip of up to 64 output rows, and walks
// down the input rows with one channel per lane (so every load is a coalesced 64-channel row of
// one pixel and nothing goes through LDS).  Per input row it takes the 21-tap HORIZONTAL sums of
// the five products from 21+21 neighbour loads, pushes them into a 21-row ring held in registers
// (105 VGPRs), and once the ring is full takes the 21-tap VERTICAL sums -- the ring is never
// rotated: slot s of phase j gets weight g[(s-j-1) mod 21], read as 21 consecutive scalars of a
// doubled table.  That is the separable minimum of 2 x 21 x 5 FMAs per (pixel, channel).
// T, S and the gamma term follow per lane; a 64-lane butterfly sums them over channels and the
// total is banked in the accumulator of lane (row - strip start), so the strip's three output
// columns leave as plain stores after the last channel block: no atomics, deterministic.
typedef __attribute__((ext_vector_type(2))) float f32x2;

// a / b to ~1.5 ulp from v_rcp_f32 and one Newton step: 4 instructions where the IEEE sequence is ~10.
// Used for the three quotients per (output row, channel) of the window pass (b > 0 there).
__device__ inline float div_nr(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = fmaf(fmaf(-b, r, 1.f), r, r);
  return a * r;
}

// One DPP move of a float (ctrl is an instruction immediate); lanes outside row_mask get 0.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp0(float v) {
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// Sum over the 64 lanes, valid in lane 63, broadcast through an SGPR: six DPP adds (quad swaps,
// half-row and row mirrors, then the two row broadcasts) instead of six LDS-crossbar bpermutes.
__device__ inline float wave_sum(float v) {
  v += dpp0<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp0<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp0<0x141, 0xF>(v);  // row_half_mirror
  v += dpp0<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
  v += dpp0<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp0<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Three wave sums at once (same adds in the same order as wave_sum, valid in lane 63).  Written as one block of
// v_add_f32_dpp: left to hipcc each step is a zeroing move + a DPP move + an add, the three chains run one after
// the other and every DPP waits two states on the add before it (~42 vector instructions and ~15 s_nop per
// output row); interleaved, the other two chains ARE the two wait states (18 instructions, no s_nop).
__device__ inline void wave_sum3(float &a, float &b, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass would hand the block to the x86 assembler)
#define NQA_DPP3(CTRL, TAIL)                                        \
  "v_add_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf" TAIL "\n\t"     \
  "v_add_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf" TAIL "\n\t"     \
  "v_add_f32_dpp %2, %2, %2 " CTRL " bank_mask:0xf" TAIL "\n\t"
  asm volatile("s_nop 1\n\t"  // the operands may come straight out of the preceding vector instruction
               NQA_DPP3("quad_perm:[1,0,3,2] row_mask:0xf", " bound_ctrl:1")
               NQA_DPP3("quad_perm:[2,3,0,1] row_mask:0xf", " bound_ctrl:1")
               NQA_DPP3("row_half_mirror row_mask:0xf", " bound_ctrl:1")
               NQA_DPP3("row_mirror row_mask:0xf", " bound_ctrl:1")
               NQA_DPP3("row_bcast:15 row_mask:0xa", "")
               NQA_DPP3("row_bcast:31 row_mask:0xc", "")
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c));
#undef NQA_DPP3
  a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
  b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
  c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
#endif
}

template <typename P, int C>  // C (64..512) is a template parameter so tap strides fold into load immediates
__global__ __launch_bounds__(256, 2) void adists_window_lanes_kernel(
    const typename P::T *__restrict__ fx, const typename P::T *__restrict__ fy, int H, int W,
    const float *__restrict__ q, int B, int ctot, int coff, const float *__restrict__ wgt, Gauss gw,
    float *__restrict__ gamma, float *__restrict__ tw, float *__restrict__ sw) {
  typedef typename P::T T;
  // the wave index as a provably uniform value, so the row pointers and loop bounds stay scalar
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int Ho = H - (kWin - 1), Wo = W - (kWin - 1);
  const int ox = blockIdx.x * 4 + wave;  // 4 adjacent columns per block: their neighbour loads overlap in L1
  if (ox >= Wo) return;
  const int oy0 = blockIdx.y * 64;
  const int nout = min(64, Ho - oy0);
  const int nrows = nout + kWin - 1;
  const size_t st = (size_t)B * ctot, qo = (size_t)b * ctot + coff;
  float acc_g = 0.f, acc_t = 0.f, acc_s = 0.f;  // lane l: output row oy0 + l
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + lane;
    const float ix = q[0 * st + qo + c], iy = q[1 * st + qo + c], wc = wgt[qo + c];
    const T *px = fx + ((size_t)(b * H + oy0) * W + ox) * C + cb;  // wave-uniform; the lane is added per load
    const T *py = fy + ((size_t)(b * H + oy0) * W + ox) * C + cb;
    // ring of the last 21 rows' horizontal sums, slot = grp*7 + sub (a dynamically indexed ring -- a switch
    // over 21 slots -- makes hipcc shuttle all of it through AGPRs every row, so the row loop is unrolled).
    // The five running sums travel as two float pairs + one float so that the 2 x 21 taps per
    // (pixel, channel) are packed v_pk_fma_f32 / v_pk_mul_f32: 4 + 3 instructions per tap pair.
    f32x2 r01[3][7], r23[3][7];
    float r4[3][7];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        r01[g][u] = r23[g][u] = (f32x2){0.f, 0.f};
        r4[g][u] = 0.f;
      }
    // 21 rows per trip, fully unrolled: the ring slot (grp*7 + sub) of every row body is a compile-time
    // constant, so a row's sums drop into their slot without selects and the vertical tap weights are
    // immediates of the kernel-argument window (no table loads)
    for (int rr0 = 0; rr0 < nrows; rr0 += 21) {
#pragma unroll
      for (int grp = 0; grp < 3; ++grp)
#pragma unroll
      for (int sub = 0; sub < 7; ++sub) {
        const int rr = rr0 + grp * 7 + sub;
        if (rr < nrows) {
          // all 42 loads go out back to back before any arithmetic (left to itself hipcc pairs each
          // load with its use and pays the memory latency 21 times per row)
          T xr[kWin], yr[kWin];
#pragma unroll
          for (int j = 0; j < kWin; ++j) {
            xr[j] = px[(size_t)j * C + lane];
            yr[j] = py[(size_t)j * C + lane];
          }
          px += (size_t)W * C;
          py += (size_t)W * C;
          __builtin_amdgcn_sched_barrier(0);
          f32x2 h01 = {0.f, 0.f}, h23 = {0.f, 0.f};
          float h4 = 0.f;
#pragma unroll
          for (int j = 0; j < kWin; ++j) {
            const f32x2 v = {P::to_f(xr[j]), P::to_f(yr[j])};
            const f32x2 gv = gw.g[j] * v;
            h01 += gv;
            h23 = gv * v + h23;
            h4 = fmaf(gv[0], v[1], h4);
          }
          r01[grp][sub] = h01;
          r23[grp][sub] = h23;
          r4[grp][sub] = h4;
          if (rr >= kWin - 1) {
            f32x2 m01 = {0.f, 0.f}, m23 = {0.f, 0.f};
            float m4 = 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int u = 0; u < 7; ++u) {
                const float wv = gw.g[(kWin - 1 - (grp * 7 + sub) + g * 7 + u) % kWin];  // slot (g,u) at this phase
                m01 = wv * r01[g][u] + m01;
                m23 = wv * r23[g][u] + m23;
                m4 = fmaf(wv, r4[g][u], m4);
              }
            const float m0 = m01[0], m1 = m01[1], m2 = m23[0], m3 = m23[1];
            const float gterm = div_nr(m2 - m0 * m0, m0 + 1e-12f);
            const float mx = ix * m0, my = iy * m1;
            const float vx = ix * ix * m2 - mx * mx, vy = iy * iy * m3 - my * my;
            const float cov = ix * iy * m4 - mx * my;
            const float tt = wc * div_nr(2.f * mx * my + 1e-6f, mx * mx + my * my + 1e-6f);
            const float ss = wc * div_nr(2.f * cov + 1e-6f, vx + vy + 1e-6f);
            const float gs = wave_sum(gterm), ts = wave_sum(tt), sss = wave_sum(ss);
            if (lane == rr - (kWin - 1)) {
              acc_g += gs;
              acc_t += ts;
              acc_s += sss;
            }
          }
        }
      }
    }
  }
  if (lane < nout) {
    const size_t o = ((size_t)b * Ho + oy0 + lane) * Wo + ox;
    gamma[o] = acc_g / (float)C;
    tw[o] = acc_t;
    sw[o] = acc_s;
  }
}

// gaussian(21, 7) as ADISTS.py:102-104 builds it (float32(exp(.)) / float32 sum), as compile-time constants:
// make_gauss() recomputes the taps at run time and adists_run refuses to start if the two disagree.
// As literals the taps are instruction immediates (v_fmac_f32 / v_mul_f32 with a 32-bit constant): no scalar
// registers, no broadcast moves -- the packed-float form above kept them in SGPRs, spilled 40-80 of them and
// built a {w, w} register pair per tap for every v_pk_fma_f32, which on this chip issues at half the rate of
// v_fma_f32 anyway.
static constexpr float kG[kWin] = {
    0x1.8453aep-6f, 0x1.d76892p-6f, 0x1.185a34p-5f, 0x1.46b8bap-5f, 0x1.75117ap-5f, 0x1.a16246p-5f, 0x1.c987c2p-5f,
    0x1.eb6810p-5f, 0x1.02907ep-4f, 0x1.0a9a20p-4f, 0x1.0d5620p-4f, 0x1.0a9a20p-4f, 0x1.02907ep-4f, 0x1.eb6810p-5f,
    0x1.c987c2p-5f, 0x1.a16246p-5f, 0x1.75117ap-5f, 0x1.46b8bap-5f, 0x1.185a34p-5f, 0x1.d76892p-6f, 0x1.8453aep-6f};

// the last 21 rows' horizontal sums of the five products, one ring slot per row
struct WinRing {
  float s0[kWin], s1[kWin], s2[kWin], s3[kWin], s4[kWin];
};

// One input row of the separable window: horizontal 21-tap sums of (x, y, x^2, y^2, xy) into ring slot SLOT
// (a compile-time constant, so the ring stays in registers and is never rotated); when `full`, the vertical
// 21-tap sums over the ring -- slot s carries weight kG[(20 - SLOT + s) mod 21] at this phase -- and from them
// the gamma term (ADISTS.py:84-86), T and S (:182-183) of this lane's channel.
template <int SLOT>
__device__ inline void win_row(const float (&x)[kWin], const float (&y)[kWin], WinRing &rg, bool full, float ix,
                               float iy, float wc, float &gterm, float &tt, float &ss) {
  // the window is symmetric (kG[j] == kG[20-j]): the centre tap starts the five sums (no zero + fma), every
  // mirrored pair of taps adds its two samples / squares / products first and takes ONE weighted fma per sum:
  // 13 operations per pair where tap-by-tap took 14, and no accumulator initialisation
  constexpr int MID = kWin / 2;
  static_assert(kWin == 21 && kG[0] == kG[20] && kG[3] == kG[17] && kG[9] == kG[11], "the window must be symmetric");
  float h0, h1, h2, h3, h4;
  {
    const float gx = kG[MID] * x[MID], gy = kG[MID] * y[MID];
    h0 = gx;
    h1 = gy;
    h2 = gx * x[MID];
    h3 = gy * y[MID];
    h4 = gx * y[MID];
  }
#pragma unroll
  for (int j = 0; j < MID; ++j) {
    const int k = kWin - 1 - j;
    const float sx = x[j] + x[k], sy = y[j] + y[k];
    const float uu = fmaf(x[k], x[k], x[j] * x[j]), vv = fmaf(y[k], y[k], y[j] * y[j]);
    const float ww = fmaf(x[k], y[k], x[j] * y[j]);
    h0 = fmaf(kG[j], sx, h0);
    h1 = fmaf(kG[j], sy, h1);
    h2 = fmaf(kG[j], uu, h2);
    h3 = fmaf(kG[j], vv, h3);
    h4 = fmaf(kG[j], ww, h4);
  }
  rg.s0[SLOT] = h0;
  rg.s1[SLOT] = h1;
  rg.s2[SLOT] = h2;
  rg.s3[SLOT] = h3;
  rg.s4[SLOT] = h4;
  if (full) {
    // slot 0 starts the vertical sums (a product, not zero + fma)
    constexpr float w0 = kG[(kWin - 1 - SLOT) % kWin];
    float m0 = w0 * rg.s0[0], m1 = w0 * rg.s1[0], m2 = w0 * rg.s2[0], m3 = w0 * rg.s3[0], m4 = w0 * rg.s4[0];
#pragma unroll
    for (int sl = 1; sl < kWin; ++sl) {
      const float wv = kG[(kWin - 1 - SLOT + sl) % kWin];
      m0 = fmaf(wv, rg.s0[sl], m0);
      m1 = fmaf(wv, rg.s1[sl], m1);
      m2 = fmaf(wv, rg.s2[sl], m2);
      m3 = fmaf(wv, rg.s3[sl], m3);
      m4 = fmaf(wv, rg.s4[sl], m4);
    }
    gterm = div_nr(m2 - m0 * m0, m0 + 1e-12f);
    const float mx = ix * m0, my = iy * m1;
    const float vx = ix * ix * m2 - mx * mx, vy = iy * iy * m3 - my * my;
    const float cov = ix * iy * m4 - mx * my;
    tt = wc * div_nr(2.f * mx * my + 1e-6f, mx * mx + my * my + 1e-6f);
    ss = wc * div_nr(2.f * cov + 1e-6f, vx + vy + 1e-6f);
  }
}

// The same pass with the taps shared through LDS (the shipped form).  The kernel above lets every wave fetch its
// own 2 x 21 neighbour pixels per input row from L1: 42 wave-wide loads per output row, of which a block's four
// adjacent columns share 21 of every 24 pixels -- measured L1/TA-bound, and with workgroups dealt round-robin
// over the 8 XCDs every pixel was fetched into up to six L2s (rocprofv3: 32 GB fetched for a 4.2 GB tap,
// L2 hit rate 0.43, the dispatch HBM-bound at 5.6 TB/s).  Here
//   * workgroup ids are remapped so that the ids that share an XCD (equal mod 8) own a contiguous run of
//     column groups: neighbouring columns' rows hit one L2 (speed only; any placement is correct);
//   * a block (4 waves = 4 adjacent output columns) brings each input row's 24-pixel window of both images
//     into LDS ONCE by LDS-DMA (1-KB pieces, 3 per wave and row instead of 42 loads), three rows ahead of
//     their use in a four-slot ring, retired by a counted vmcnt + one barrier per row;
//   * the 42 taps of a wave are ds_read_b32 at immediate offsets from one address register.
// The arithmetic (horizontal sums, 21-row register ring, vertical sums, T / S / gamma, channel reduction) is
// the kernel above, instruction for instruction.
typedef __attribute__((address_space(3))) void lds_void_a_t;

template <typename P, int C>
__global__ __launch_bounds__(256, 3) void adists_window_lds_kernel(
    const typename P::T *__restrict__ fx, const typename P::T *__restrict__ fy, int H, int W,
    const float *__restrict__ q, int B, int ctot, int coff, const float *__restrict__ wgt, Gauss gw,
    float *__restrict__ gamma, float *__restrict__ tw, float *__restrict__ sw, int nbx, int nby, int strip) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)gw;  // the taps are the compile-time constants kG (checked against make_gauss() on the host)
  typedef typename P::T T;
  constexpr int SZ = (int)sizeof(T);
  constexpr int PXB = 64 * SZ;              // one pixel's 64-channel block, bytes
  constexpr int PPP = 1024 / PXB;           // pixels per 1-KB DMA piece: 4 (float) or 8 (16-bit)
  constexpr int PPI = SZ == 4 ? 6 : 4;      // pieces per image and row: 24 pixels (float), padded to 32 (16-bit)
  constexpr int WPX = PPI * PPP;            // window pixels held per image
  constexpr int NPW = 2 * PPI / 4;          // pieces per wave and row: 3 or 2
  constexpr int ROWB = 2 * WPX * PXB;       // one ring slot: both images' window of one input row
  constexpr int D = 3, R = D + 1;           // rows in flight ahead of the one being read; ring slots
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware decode of the linear workgroup id -> (column group, row strip, image pair)
  int id = blockIdx.x;
  {
    const int nb = gridDim.x, qq = nb >> 3, rr8 = nb & 7, xcd = id & 7, local = id >> 3;
    id = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + local;
  }
  const int bx = id % nbx, by = (id / nbx) % nby, b = id / (nbx * nby);
  const int Ho = H - (kWin - 1), Wo = W - (kWin - 1);
  const int ox0 = bx * 4, ox = ox0 + wave;
  const bool live_col = ox < Wo;            // a dead column computes on clamped pixels and stores nothing
  // a block walks a strip of `strip` output rows (the launcher's choice): the 20 rows of vertical run-in are
  // paid once per strip, so tall strips where the grid stays large enough (64-row strips: 31 % more input rows)
  const int oy0 = by * strip;
  const int nout = min(strip, Ho - oy0);
  const int nrows = nout + kWin - 1;
  const size_t st = (size_t)B * ctot, qo = (size_t)b * ctot + coff;
  const unsigned img_bytes = (unsigned)H * (unsigned)W * (unsigned)C * (unsigned)SZ;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T *>(fx + (size_t)b * H * W * C), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T *>(fy + (size_t)b * H * W * C), 0, img_bytes, 0x00020000);
  // this wave's DMA pieces: piece p = i*4 + wave covers pixels [pidx*PPP, +PPP) of image p / PPI
  unsigned voff[NPW];  // per-lane source offset of piece i at row 0 of the strip, channel block 0
  int pdst[NPW], pimg[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pc = i * 4 + wave;
    pimg[i] = pc / PPI;
    const int pidx = pc - pimg[i] * PPI;
    pdst[i] = pimg[i] * (WPX * PXB) + pidx * 1024;
    const int px = min(ox0 + pidx * PPP + lane / (64 / PPP), W - 1);
    voff[i] = (unsigned)(((oy0 * W + px) * C) * SZ + (lane % (64 / PPP)) * 16);
  }
  const unsigned row_stride = (unsigned)W * (unsigned)C * (unsigned)SZ;
  const int rd_base = wave * PXB + lane * SZ;  // this lane's tap 0 of image x inside a slot
  float acc_g = 0.f, acc_t = 0.f, acc_s = 0.f;  // lane l: output row oy0 + 64 k + l of the current 64-row group k
  // (Moving two of these accumulators to LDS for C > 64 was tried to spare registers: it takes the block's LDS past a
  // third of the CU's -- 2 blocks per CU instead of 3 -- see profiles/r03_window_ab.txt.)
  for (int cb = 0; cb < C; cb += 64) {
    const int c = cb + lane;
    // a finished group of (up to) 64 output rows leaves the lanes: stored by the first channel block, added to
    // by the later ones (same lane, same address, program order), the last one scales gamma by 1/C
    auto flush = [&](int orow) {
      const int r = (orow & ~63) + lane;
      if (live_col && r <= orow) {
        const size_t o = ((size_t)b * Ho + oy0 + r) * Wo + ox;
        float fg = acc_g, ft = acc_t, fs = acc_s;
        if (cb > 0) {
          fg += gamma[o];
          ft += tw[o];
          fs += sw[o];
        }
        gamma[o] = cb + 64 >= C ? fg / (float)C : fg;
        tw[o] = ft;
        sw[o] = fs;
      }
      acc_g = acc_t = acc_s = 0.f;
    };
    // the channel's three constants live in LDS (re-read per output row at immediate offsets): with them in
    // registers the C > 64 instances need 172 VGPRs, four past the three-waves-per-SIMD limit
    float *const chan = reinterpret_cast<float *>(smem_w + R * ROWB) + wave * 192 + lane;
    chan[0] = q[0 * st + qo + c];
    chan[64] = q[1 * st + qo + c];
    chan[128] = wgt[qo + c];
    auto issue_row = [&](int r) {
      char *slot = smem_w + (r % R) * ROWB;
#pragma unroll
      for (int i = 0; i < NPW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pimg[i] ? ry : rx, (lds_void_a_t *)(slot + pdst[i]), 16,
                                                 voff[i] + (unsigned)(cb * SZ), (unsigned)r * row_stride, 0, 0);
    };
    // every wave has finished reading the previous channel block's last rows before their slots are refilled
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < D; ++r)
      if (r < nrows) issue_row(r);
    WinRing rg;
#pragma unroll
    for (int u = 0; u < kWin; ++u) rg.s0[u] = rg.s1[u] = rg.s2[u] = rg.s3[u] = rg.s4[u] = 0.f;
    // 21 rows per trip, the row body instantiated once per ring slot: every slot index and every vertical tap
    // weight is a compile-time constant
#define NQA_WIN_ROW(SLOT)                                                                                          \
  {                                                                                                                \
    const int rr = rr0 + SLOT;                                                                                     \
    if (rr < nrows) {                                                                                              \
      /* row rr has landed (this wave's pieces: all but the D-1 younger rows'; everyone's: the barrier), and */   \
      /* every wave is done with row rr-1, whose slot row rr+D now takes */                                        \
      if (rr + D <= nrows)                                                                                         \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * NPW) : "memory");               \
      else                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                   \
      if (rr + D < nrows) issue_row(rr + D);                                                                       \
      const char *slot = smem_w + (rr % R) * ROWB + rd_base;                                                       \
      float xr[kWin], yr[kWin];                                                                                    \
      _Pragma("unroll") for (int j = 0; j < kWin; ++j) {                                                           \
        xr[j] = P::to_f(*reinterpret_cast<const T *>(slot + j * PXB));                                             \
        yr[j] = P::to_f(*reinterpret_cast<const T *>(slot + WPX * PXB + j * PXB));                                 \
      }                                                                                                            \
      float gterm = 0.f, tt = 0.f, ss = 0.f;                                                                       \
      const bool full = rr >= kWin - 1;                                                                            \
      win_row<SLOT>(xr, yr, rg, full, chan[0], chan[64], chan[128], gterm, tt, ss);                                \
      if (full) {                                                                                                  \
        float gs = gterm, ts = tt, sss = ss;                                                                       \
        wave_sum3(gs, ts, sss);                                                                                    \
        const int orow = rr - (kWin - 1);                                                                          \
        if (lane == (orow & 63)) {                                                                                 \
          acc_g += gs;                                                                                             \
          acc_t += ts;                                                                                             \
          acc_s += sss;                                                                                            \
        }                                                                                                          \
        if ((orow & 63) == 63 || orow == nout - 1) flush(orow);                                                    \
      }                                                                                                            \
    }                                                                                                              \
  }
    for (int rr0 = 0; rr0 < nrows; rr0 += kWin) {
      NQA_WIN_ROW(0) NQA_WIN_ROW(1) NQA_WIN_ROW(2) NQA_WIN_ROW(3) NQA_WIN_ROW(4) NQA_WIN_ROW(5) NQA_WIN_ROW(6)
      NQA_WIN_ROW(7) NQA_WIN_ROW(8) NQA_WIN_ROW(9) NQA_WIN_ROW(10) NQA_WIN_ROW(11) NQA_WIN_ROW(12) NQA_WIN_ROW(13)
      NQA_WIN_ROW(14) NQA_WIN_ROW(15) NQA_WIN_ROW(16) NQA_WIN_ROW(17) NQA_WIN_ROW(18) NQA_WIN_ROW(19) NQA_WIN_ROW(20)
    }
#undef NQA_WIN_ROW
  }
#endif
}

// The windowed pass of stage 0 (the raw image: 3 float planes, NCHW) in the same shape as the lanes
// kernel above, with lanes = 64 adjacent output COLUMNS of one plane: every tap load is a coalesced
// row segment at an immediate offset from one row pointer, the ring and the packed tap arithmetic
// are identical, nothing is reduced across lanes, and the three channels are accumulated into the
// maps one after the other by the same lane (plain read-modify-write, no atomics).
__global__ __launch_bounds__(256, 2) void adists_window_planar_kernel(
    const float *__restrict__ x, const float *__restrict__ y, int H, int W, const float *__restrict__ q, int B,
    int ctot, int coff, const float *__restrict__ wgt, Gauss gw, float *__restrict__ gamma, float *__restrict__ tw,
    float *__restrict__ sw) {
  (void)gw;  // taps = the compile-time constants kG
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int Ho = H - (kWin - 1), Wo = W - (kWin - 1);
  const int ox0 = blockIdx.x * 256 + wave * 64;
  if (ox0 >= Wo) return;
  const int ox = ox0 + lane;
  const bool live = ox < Wo;
  const int oxc = live ? ox : Wo - 1;  // dead lanes re-read the last column and store nothing
  const int oy0 = blockIdx.y * 64;
  const int nout = min(64, Ho - oy0);
  const int nrows = nout + kWin - 1;
  const size_t st = (size_t)B * ctot, qo = (size_t)b * ctot + coff;
  const size_t o0 = ((size_t)b * Ho + oy0) * Wo + ox;
  for (int c = 0; c < 3; ++c) {
    const float ix = q[0 * st + qo + c], iy = q[1 * st + qo + c], wc = wgt[qo + c];
    const float *px = x + ((size_t)(b * 3 + c) * H + oy0) * W + oxc;
    const float *py = y + ((size_t)(b * 3 + c) * H + oy0) * W + oxc;
    WinRing rg;
#pragma unroll
    for (int u = 0; u < kWin; ++u) rg.s0[u] = rg.s1[u] = rg.s2[u] = rg.s3[u] = rg.s4[u] = 0.f;
#define NQA_PLANAR_ROW(SLOT)                                                          \
  {                                                                                   \
    const int rr = rr0 + SLOT;                                                        \
    if (rr < nrows) {                                                                 \
      float xr[kWin], yr[kWin];                                                       \
      _Pragma("unroll") for (int j = 0; j < kWin; ++j) {                              \
        xr[j] = px[j];                                                                \
        yr[j] = py[j];                                                                \
      }                                                                               \
      px += W;                                                                        \
      py += W;                                                                        \
      float gterm = 0.f, tt = 0.f, ss = 0.f;                                          \
      const bool full = rr >= kWin - 1;                                               \
      win_row<SLOT>(xr, yr, rg, full, ix, iy, wc, gterm, tt, ss);                     \
      if (full && live) {                                                             \
        const size_t o = o0 + (size_t)(rr - (kWin - 1)) * Wo;                         \
        if (c == 0) {                                                                 \
          gamma[o] = gterm;                                                           \
          tw[o] = tt;                                                                 \
          sw[o] = ss;                                                                 \
        } else {                                                                      \
          const float gsum = gamma[o] + gterm;                                        \
          gamma[o] = c == 2 ? gsum / 3.f : gsum;                                      \
          tw[o] += tt;                                                                \
          sw[o] += ss;                                                                \
        }                                                                             \
      }                                                                               \
    }                                                                                 \
  }
    for (int rr0 = 0; rr0 < nrows; rr0 += kWin) {
      NQA_PLANAR_ROW(0) NQA_PLANAR_ROW(1) NQA_PLANAR_ROW(2) NQA_PLANAR_ROW(3) NQA_PLANAR_ROW(4) NQA_PLANAR_ROW(5)
      NQA_PLANAR_ROW(6) NQA_PLANAR_ROW(7) NQA_PLANAR_ROW(8) NQA_PLANAR_ROW(9) NQA_PLANAR_ROW(10) NQA_PLANAR_ROW(11)
      NQA_PLANAR_ROW(12) NQA_PLANAR_ROW(13) NQA_PLANAR_ROW(14) NQA_PLANAR_ROW(15) NQA_PLANAR_ROW(16) NQA_PLANAR_ROW(17)
      NQA_PLANAR_ROW(18) NQA_PLANAR_ROW(19) NQA_PLANAR_ROW(20)
    }
#undef NQA_PLANAR_ROW
  }
}

// Global branch of one stage (maps smaller than the window; ADISTS.py:91-97,176-180): one
// block per image pair, from the per-channel global statistics.  Outputs 1x1 "maps".
__global__ __launch_bounds__(256) void adists_global_kernel(const float *__restrict__ q, int B, int ctot, int coff,
                                                            int creal, const float *__restrict__ wgt,
                                                            float *__restrict__ gamma, float *__restrict__ tw,
                                                            float *__restrict__ sw) {
  __shared__ double red[3][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t st = (size_t)B * ctot, qo = (size_t)b * ctot + coff;
  double g = 0, t = 0, s = 0;
  for (int c = tid; c < creal; c += 256) {
    const float ix = q[0 * st + qo + c], iy = q[1 * st + qo + c], w = wgt[qo + c];
    const float rmx = q[3 * st + qo + c], rmy = q[4 * st + qo + c];
    const float rvx = q[5 * st + qo + c], rvy = q[6 * st + qo + c], rcov = q[7 * st + qo + c];
    g += (double)(rvx / (rmx + 1e-12f));
    const float mx = ix * rmx, my = iy * rmy;
    const float vx = ix * ix * rvx, vy = iy * iy * rvy, cov = ix * iy * rcov;
    t += (double)(w * ((2.f * mx * my + 1e-6f) / (mx * mx + my * my + 1e-6f)));
    s += (double)(w * ((2.f * cov + 1e-6f) / (vx + vy + 1e-6f)));
  }
  red[0][tid] = g;
  red[1][tid] = t;
  red[2][tid] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off)
      for (int j = 0; j < 3; ++j) red[j][tid] += red[j][tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    gamma[b] = (float)(red[0][0] / creal);
    tw[b] = (float)red[1][0];
    sw[b] = (float)red[2][0];
  }
}

// ---------------------------------------------------------------------------------
// Probability chain on the small maps (compute_prob, ADISTS.py:77-99) and the stage's D.
struct ChainAcc {       // one per (stage, image); zeroed before the chain
  double sum, sumsq;    // of gamma
  unsigned ps_min, ps_max, pp_min, pp_max;  // float bit patterns (values >= 0)
  double dsum;          // sum over the map of (1-ps) TW + ps SW
};

__device__ inline float sigmoid_ref(float z) { return 1.f / (1.f + expf(-z)); }

__device__ inline void zscore(const ChainAcc &a, int n, float &mean, float &sd) {
  const double m = a.sum / n;
  mean = (float)m;
  double v = (a.sumsq - n * m * m) / (double)(n - 1);  // unbiased; n==1 -> 0/0 = NaN as torch.std
  if (v < 0.0) v = 0.0;                                  // rounding of a constant map (NaN stays NaN)
  sd = (float)sqrt(v);
}

__global__ __launch_bounds__(256) void chain_init_kernel(ChainAcc *acc, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    acc[i].sum = acc[i].sumsq = acc[i].dsum = 0.0;
    acc[i].ps_min = acc[i].pp_min = 0x7F800000u;  // +inf
    acc[i].ps_max = acc[i].pp_max = 0u;
  }
}

// Cross-block sums of the chain (gamma moments, the stage's D) are NOT accumulated with fp64 atomics: their
// arrival order would make the last bits of a score differ from run to run.  Each block stores its partial
// in a slot of its own and chain_fold_kernel adds the slots in a fixed order (min / max stay atomic: exact).
__global__ __launch_bounds__(256) void chain_fold_kernel(const double *__restrict__ cp, int G, int nvals, int which,
                                                         ChainAcc *acc) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int v = 0; v < nvals; ++v) {
    double s = 0.0;
    for (int g = tid; g < G; g += 256) s += cp[((size_t)b * G + g) * nvals + v];
    __syncthreads();
    red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) {
      if (which == 1) acc[b].dsum = red[0];
      else if (v == 0) acc[b].sum = red[0];
      else acc[b].sumsq = red[0];
    }
  }
}

__global__ __launch_bounds__(256) void chain_moments_kernel(const float *__restrict__ gamma, int n,
                                                            double *__restrict__ cp /* [B][gridDim.x][2] */) {
  __shared__ double red[2][256];
  const int b = blockIdx.y, tid = threadIdx.x;
  double s = 0, s2 = 0;
  for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
    const double g = gamma[(size_t)b * n + i];
    s += g;
    s2 += g * g;
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      red[0][tid] += red[0][tid + off];
      red[1][tid] += red[1][tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    cp[((size_t)b * gridDim.x + blockIdx.x) * 2 + 0] = red[0][0];
    cp[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

__device__ inline void block_minmax(float lo, float hi, unsigned *gmin, unsigned *gmax) {
  __shared__ float rl[256], rh[256];
  const int tid = threadIdx.x;
  rl[tid] = lo;
  rh[tid] = hi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      rl[tid] = fminf(rl[tid], rl[tid + off]);
      rh[tid] = fmaxf(rh[tid], rh[tid + off]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    atomicMin(gmin, __float_as_uint(rl[0]));
    atomicMax(gmax, __float_as_uint(rh[0]));
  }
}

__global__ __launch_bounds__(256) void chain_psminmax_kernel(const float *__restrict__ gamma, int n, ChainAcc *acc) {
  const int b = blockIdx.y;
  float mean, sd;
  zscore(acc[b], n, mean, sd);
  float lo = INFINITY, hi = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float ps = sigmoid_ref((gamma[(size_t)b * n + i] - mean) / (sd + 1e-12f));
    lo = fminf(lo, ps);
    hi = fmaxf(hi, ps);
  }
  block_minmax(lo, hi, &acc[b].ps_min, &acc[b].ps_max);
}

// bilinear sample of prev (hp x wp) at output (y,x) of an (h x w) grid, align_corners=True
__device__ inline float bilinear_ac(const float *__restrict__ prev, int hp, int wp, int h, int w, int y, int x) {
  const float sy = h > 1 ? (float)(hp - 1) / (float)(h - 1) : 0.f;
  const float sx = w > 1 ? (float)(wp - 1) / (float)(w - 1) : 0.f;
  const float ry = sy * y, rx = sx * x;
  const int y0 = (int)ry, x0 = (int)rx;
  const int y1 = y0 + (y0 < hp - 1 ? 1 : 0), x1 = x0 + (x0 < wp - 1 ? 1 : 0);
  const float ly = ry - y0, lx = rx - x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * prev[y0 * wp + x0] + lx * prev[y0 * wp + x1]) +
         ly * (hx * prev[y1 * wp + x0] + lx * prev[y1 * wp + x1]);
}

__device__ inline float pp_value(const float *gamma, const ChainAcc &a, int n, const float *prev, int hp, int wp,
                                 int h, int w, int i, float mean, float sd) {
  const float ps = sigmoid_ref((gamma[i] - mean) / (sd + 1e-12f));
  const float lo = __uint_as_float(a.ps_min), hi = __uint_as_float(a.ps_max);
  const float psn = (ps - lo) / (hi - lo + 1e-12f);
  return psn * bilinear_ac(prev, hp, wp, h, w, i / w, i % w);
}

__global__ __launch_bounds__(256) void chain_ppminmax_kernel(const float *__restrict__ gamma, int h, int w,
                                                             const float *__restrict__ prev, int hp, int wp,
                                                             ChainAcc *acc) {
  const int b = blockIdx.y, n = h * w;
  float mean, sd;
  zscore(acc[b], n, mean, sd);
  float lo = INFINITY, hi = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float pp = pp_value(gamma + (size_t)b * n, acc[b], n, prev + (size_t)b * hp * wp, hp, wp, h, w, i, mean, sd);
    lo = fminf(lo, pp);
    hi = fmaxf(hi, pp);
  }
  block_minmax(lo, hi, &acc[b].pp_min, &acc[b].pp_max);
}

__global__ __launch_bounds__(256) void chain_final_kernel(const float *__restrict__ gamma, int h, int w,
                                                          const float *__restrict__ prev, int hp, int wp,
                                                          const float *__restrict__ tw, const float *__restrict__ sw,
                                                          float *__restrict__ psprod, const ChainAcc *acc,
                                                          double *__restrict__ cp /* [B][gridDim.x] */) {
  __shared__ double red[256];
  const int b = blockIdx.y, n = h * w, tid = threadIdx.x;
  float mean, sd;
  zscore(acc[b], n, mean, sd);
  const float lo = __uint_as_float(acc[b].pp_min), hi = __uint_as_float(acc[b].pp_max);
  double d = 0.0;
  for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
    const float pp = pp_value(gamma + (size_t)b * n, acc[b], n, prev + (size_t)b * hp * wp, hp, wp, h, w, i, mean, sd);
    const float ps = (pp - lo) / (hi - lo + 1e-12f);
    psprod[(size_t)b * n + i] = ps;
    d += (double)((1.f - ps) * tw[(size_t)b * n + i] + ps * sw[(size_t)b * n + i]);
  }
  red[tid] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) cp[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

// global-branch stage: ps = sigmoid(gamma); ps_prod = ps * prev[0,0]; D = (1-ps_prod) TW + ps_prod SW
__global__ void chain_global_kernel(const float *__restrict__ gamma, const float *__restrict__ prev, int hpwp,
                                    const float *__restrict__ tw, const float *__restrict__ sw,
                                    float *__restrict__ psprod, ChainAcc *acc, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float ps = sigmoid_ref(gamma[b]) * prev[(size_t)b * hpwp];
  psprod[b] = ps;
  acc[b].dsum = (double)((1.f - ps) * tw[b] + ps * sw[b]);
}

__global__ void ones_kernel(float *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1.f;
}

// D_b = sum_k dsum_k / n_k
struct DDesc {
  int n[NQA_NUM_TAPS];
};
__global__ void adists_d_kernel(const ChainAcc *acc, DDesc d, int B, float *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int k = NQA_NUM_TAPS - 1; k >= 0; --k) s += (float)(acc[(size_t)k * B + b].dsum / d.n[k]);
  out[b] = s;
}

// as_map=True (ADISTS.py:163,188-189,193): 1 - sum over stages (coarse to fine, the reference's loop
// order) of the stage's D map bilinearly resized to (H, W), align_corners=False.  The stage map is
// rebuilt from the chain's ps_prod and the window pass' TW / SW maps.
struct MapDesc {
  const float *ps[NQA_NUM_TAPS], *tw[NQA_NUM_TAPS], *sw[NQA_NUM_TAPS];
  int mh[NQA_NUM_TAPS], mw[NQA_NUM_TAPS];
};
__device__ inline void src_index(int dst, int in, int out, int &i0, int &i1, float &l0, float &l1) {
  // torch upsample_bilinear2d, align_corners=False: src = in/out * (dst + 0.5) - 0.5, clamped at 0
  const float scale = (float)in / (float)out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void adists_map_kernel(MapDesc d, int H, int W, float *__restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  float s = 0.f;
  for (int k = NQA_NUM_TAPS - 1; k >= 0; --k) {
    const int mh = d.mh[k], mw = d.mw[k];
    const size_t o = (size_t)b * mh * mw;
    const float *ps = d.ps[k] + o, *tw = d.tw[k] + o, *sw = d.sw[k] + o;
    auto dm = [&](int yy, int xx) {
      const int j = yy * mw + xx;
      const float p = ps[j];
      return (1.f - p) * tw[j] + p * sw[j];
    };
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(y, mh, H, y0, y1, ly0, ly1);
    src_index(x, mw, W, x0, x1, lx0, lx1);
    s = s + (ly0 * (lx0 * dm(y0, x0) + lx1 * dm(y0, x1)) + ly1 * (lx0 * dm(y1, x0) + lx1 * dm(y1, x1)));
  }
  out[(size_t)b * H * W + i] = 1.f - s;
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static Gauss make_gauss() {
  // gaussian(21, 7): exp(-(x-10)^2 / (2*7^2)) in double, stored as float, normalised in float
  // (ADISTS.py:102-104 builds a float32 tensor and divides by its float32 sum)
  Gauss g;
  float s = 0.f;
  float v[kWin];
  for (int i = 0; i < kWin; ++i) v[i] = (float)exp(-(double)((i - 10) * (i - 10)) / (2.0 * 7.0 * 7.0));
  // torch.sum of 21 floats: sequential order is within 1 ulp of any other; use double then round
  double sd = 0.0;
  for (int i = 0; i < kWin; ++i) sd += (double)v[i];
  s = (float)sd;
  for (int i = 0; i < kWin; ++i) g.g[i] = v[i] / s;
  return g;
}

struct APlan {
  // byte offsets into the workspace
  size_t bufA, bufB, taps[5], img4x, img4y, part, q, ent, wgt, maps[NQA_NUM_TAPS][4], acc, chain_part, total;
  StageDesc sd;
  EntDesc ed;
  int h[NQA_NUM_TAPS], w[NQA_NUM_TAPS], c[NQA_NUM_TAPS];  // feature dims per tap (k=0 raw image)
  int mh[NQA_NUM_TAPS], mw[NQA_NUM_TAPS];                  // map dims (1x1 for the global branch)
  bool windowed[NQA_NUM_TAPS];
  int ent_ppb[NQA_NUM_TAPS];
};

static APlan make_plan(int B, int H, int W, int prec) {
  APlan p;
  memset(&p, 0, sizeof(p));
  const size_t esz = prec_elem_bytes(prec);
  p.h[0] = H;
  p.w[0] = W;
  p.c[0] = 3;
  int h = H, w = W;
  for (int k = 1; k < 6; ++k) {
    if (k > 1) {
      h = (h + 1) / 2;
      w = (w + 1) / 2;
    }
    p.h[k] = h;
    p.w[k] = w;
    p.c[k] = kChns[k];
  }
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align_up(bytes, 256);
    return o;
  };
  const size_t act = (size_t)2 * B * max_act_elems(H, W) * esz;  // not H*W*64: tiny frames peak at a later stage
  p.bufA = take(act);
  p.bufB = take(act);
  for (int k = 0; k < 5; ++k) p.taps[k] = take((size_t)2 * B * p.h[k + 1] * p.w[k + 1] * p.c[k + 1] * esz);
  p.img4x = take((size_t)B * H * W * 4 * 4);
  p.img4y = take((size_t)B * H * W * 4 * 4);
  // statistics partials: stage 0 from the NCHW kernel, stages 1..5 from the NHWC kernel
  long doff = 0;
  int coff = 0;
  for (int k = 0; k < 6; ++k) {
    const int hw = p.h[k] * p.w[k];
    // taps 1..4 get their statistics inside the fused pool pass (items = pooled pixels)
    int nblk;
    if (k == 0)
      nblk = cdiv(hw, stats_nchw_ppb(hw));
    else if (k <= 4)
      nblk = pool_stats_tiles((p.h[k] + 1) / 2, (p.w[k] + 1) / 2, p.c[k], prec, B, nullptr, nullptr);
    else
      nblk = cdiv(hw, stats_units_per_block(hw, p.c[k], prec, B));
    p.sd.part_off[k] = doff;
    p.sd.nblk[k] = nblk;
    p.sd.hw[k] = hw;
    p.sd.c[k] = p.c[k];
    p.sd.coff[k] = coff;
    doff += (long)B * p.sd.nblk[k] * p.c[k] * 5;
    coff += p.c[k];
  }
  p.sd.nstage = 6;
  p.sd.ctot = coff;
  p.part = take((size_t)doff * 8);
  p.q = take((size_t)8 * B * coff * 4);
  long eoff = 0;
  for (int k = 0; k < 6; ++k) {
    const int hw = p.h[k] * p.w[k];
    const int cpad = k == 0 ? 4 : p.c[k];
    const int ppb = stats_units_per_block(hw, cpad, k == 0 ? NQA_PREC_F32 : prec, B);
    p.ent_ppb[k] = ppb;
    p.ed.part_off[k] = eoff;
    p.ed.nblk[k] = cdiv(hw, ppb);
    p.ed.c[k] = cpad;
    p.ed.creal[k] = p.c[k];
    p.ed.coff[k] = p.sd.coff[k];
    eoff += (long)B * p.ed.nblk[k] * cpad;
  }
  p.ed.ctot = coff;
  p.ent = take((size_t)eoff * 8);
  p.wgt = take((size_t)B * coff * 4);
  for (int k = 0; k < 6; ++k) {
    p.windowed[k] = p.h[k] >= kWin && p.w[k] >= kWin;
    p.mh[k] = p.windowed[k] ? p.h[k] - (kWin - 1) : 1;
    p.mw[k] = p.windowed[k] ? p.w[k] - (kWin - 1) : 1;
    for (int j = 0; j < 4; ++j) p.maps[k][j] = take((size_t)B * p.mh[k] * p.mw[k] * 4);
  }
  p.acc = take((size_t)6 * B * sizeof(ChainAcc));
  p.chain_part = take((size_t)B * kChainBlocks * 2 * sizeof(double));
  p.total = off;
  return p;
}

template <typename P>
static int launch_entropy(const void *feat, int B, int HW, int C, int ppb, const float *invx, const float *sumx,
                          int ctot, double *part, hipStream_t st) {
  dim3 grid(cdiv(HW, ppb), B);
  TimedLaunch t(NQA_K_ADISTS, st);
  entropy_nhwc_kernel<P><<<grid, 256, 0, st>>>(reinterpret_cast<const typename P::T *>(feat), HW, C, ppb, invx, sumx,
                                               ctot, part);
  return check_launch("entropy");
}

static int launch_window_planar(const float *x, const float *y, int B, int H, int W, const float *q, int ctot, int coff,
                                const float *wgt, const Gauss &g, float *gamma, float *tw, float *sw, hipStream_t st) {
  const int Ho = H - (kWin - 1), Wo = W - (kWin - 1);
  dim3 grid(cdiv(Wo, 256), cdiv(Ho, 64), B);
  TimedLaunch t(NQA_K_ADISTS, st);
  adists_window_planar_kernel<<<grid, 256, 0, st>>>(x, y, H, W, q, B, ctot, coff, wgt, g, gamma, tw, sw);
  return check_launch("adists_window_planar");
}

template <typename P>
static int launch_window_lanes(const void *fx, const void *fy, int B, int H, int W, int C, const float *q, int ctot,
                               int coff, const float *wgt, const Gauss &g, float *gamma, float *tw, float *sw,
                               hipStream_t st) {
  const int Ho = H - (kWin - 1), Wo = W - (kWin - 1);
  const typename P::T *px = reinterpret_cast<const typename P::T *>(fx), *py = reinterpret_cast<const typename P::T *>(fy);
  // the shipped form for float taps (f32 / f32s): taps shared through LDS, XCD-aware column order.  16-bit taps
  // (the opt-in f16 / bf16 modes) measured 20 % slower that way (2-byte LDS reads + conversions) and keep the first form
  if constexpr (sizeof(typename P::T) == 4) {  // (the LDS kernel is instantiated for float taps only)
  if (!adists_window_legacy()) {
    // four ring slots + the waves' channel constants (3 x 64 floats each): 51 KB, three blocks per CU
    const int LDS = 4 * 2 * 24 * 64 * 4 + 4 * 768;
    // strips as tall as the grid allows (up to 256 rows): the fewest strips that still give the chip ~8 rounds
    // of blocks (3 blocks per CU), else 64-row strips
    const int nbx = cdiv(Wo, 4);
    int nby = cdiv(Ho, 256);
    while (nby < cdiv(Ho, 64) && (long)nbx * nby * B < 6144) ++nby;
    const int strip = cdiv(Ho, nby);
    nby = cdiv(Ho, strip);
    const long nblk = (long)nbx * nby * B;
    if (nblk > 0x7FFFFFFFL) {
      set_error("adists_window: grid too large");
      return NQA_E_SHAPE;
    }
    TimedLaunch t(NQA_K_ADISTS, st);
#define NQA_WIN(CC)                                                                                                  \
  adists_window_lds_kernel<P, CC><<<(unsigned)nblk, 256, LDS, st>>>(px, py, H, W, q, B, ctot, coff, wgt, g, gamma, \
                                                                    tw, sw, nbx, nby, strip)
    switch (C) {
      case 64: NQA_WIN(64); break;
      case 128: NQA_WIN(128); break;
      case 256: NQA_WIN(256); break;
      case 512: NQA_WIN(512); break;
      default: set_error("adists_window: unsupported channel count %d", C); return NQA_E_SHAPE;
    }
#undef NQA_WIN
    return check_launch("adists_window_lds");
  }
  }
  dim3 grid(cdiv(Wo, 4), cdiv(Ho, 64), B);
  TimedLaunch t(NQA_K_ADISTS, st);
  switch (C) {
    case 64: adists_window_lanes_kernel<P, 64><<<grid, 256, 0, st>>>(px, py, H, W, q, B, ctot, coff, wgt, g, gamma, tw, sw); break;
    case 128: adists_window_lanes_kernel<P, 128><<<grid, 256, 0, st>>>(px, py, H, W, q, B, ctot, coff, wgt, g, gamma, tw, sw); break;
    case 256: adists_window_lanes_kernel<P, 256><<<grid, 256, 0, st>>>(px, py, H, W, q, B, ctot, coff, wgt, g, gamma, tw, sw); break;
    case 512: adists_window_lanes_kernel<P, 512><<<grid, 256, 0, st>>>(px, py, H, W, q, B, ctot, coff, wgt, g, gamma, tw, sw); break;
    default: set_error("adists_window_lanes: unsupported channel count %d", C); return NQA_E_SHAPE;
  }
  return check_launch("adists_window_lanes");
}

}  // namespace nqa

using namespace nqa;

extern "C" {

size_t nqa_adists_workspace_bytes(int B, int H, int W, int prec) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return make_plan(B, H, W, prec).total;
}

}  // extern "C"

static int adists_run(const float *x, const float *y, int B, int H, int W, const void *packed, int prec, void *ws,
                      size_t ws_bytes, float *d_out, float *map_out, void *stream) {
  if (!x || !y || !packed || !ws || !d_out) {
    set_error("adists_forward: null pointer");
    return NQA_E_ARG;
  }
  if (B <= 0 || H <= 0 || W <= 0 || !prec_valid(prec)) {
    set_error("adists_forward: bad size or prec (B=%d H=%d W=%d prec=%d)", B, H, W, prec);
    return NQA_E_ARG;
  }
  if ((long)H * W * 64 * (long)prec_elem_bytes(prec) >= (1L << 31)) {
    set_error("adists_forward: image too large for 32-bit in-image byte offsets");
    return NQA_E_ARG;
  }
  const APlan p = make_plan(B, H, W, prec);
  if (ws_bytes < p.total) {
    set_error("adists_forward: workspace %zu < %zu bytes", ws_bytes, p.total);
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *base = static_cast<char *>(ws);
  const size_t esz = prec_elem_bytes(prec);
  const int ctot = p.sd.ctot;
  double *part = reinterpret_cast<double *>(base + p.part);
  float *q = reinterpret_cast<float *>(base + p.q);
  double *ent = reinterpret_cast<double *>(base + p.ent);
  float *wgt = reinterpret_cast<float *>(base + p.wgt);
  ChainAcc *acc = reinterpret_cast<ChainAcc *>(base + p.acc);
  void *taps[5];
  for (int k = 0; k < 5; ++k) taps[k] = base + p.taps[k];
  int rc;

  // ---- pyramids (x images [0,B), y images [B,2B)) with the global statistics per tap ----
  if ((rc = stats_nchw(x, y, B, 3, H * W, part + p.sd.part_off[0], st))) return rc;
  // (f32s: conv1_regw_split_kernel, as in nqa_api.hip's run_stages; frames in f32s are >= 128 x 128 pixels under the
  // default `auto`, far from the tiny-frame knife edge that made a conv1_1 rounding pattern matter, section 4.4)
  const bool fused_s = prec == NQA_PREC_F32S && !mixed_stage1_unfused();
  const bool fused1 = prec_elem_bytes(prec) == 2 || fused_s;
  if (!fused1) {
    if ((rc = conv1_1(x, B, H, W, packed, prec, base + p.bufA, st))) return rc;
    if ((rc = conv1_1(y, B, H, W, packed, prec, base + p.bufA + (size_t)B * H * W * 64 * esz, st))) return rc;
  }
  {
    // same chaining as nqa_vgg_pyramid, taps kept
    void *bufA = base + p.bufA, *bufB = base + p.bufB, *cur = bufA;
    for (int layer = 1; layer < NQA_NUM_CONVS; ++layer) {
      const ConvSpec &cs = kConvs[layer];
      const int k = cs.stage;
      void *dst = cs.last ? taps[k] : (cur == bufA ? bufB : bufA);
      if (layer == 1 && fused1) {
        if ((rc = fused_s ? conv1_fused_split(x, y, B, 2 * B, H, W, packed, dst, st)
                          : conv1_fused(x, y, B, 2 * B, H, W, packed, prec, dst, st)))
          return rc;
      } else if ((rc = conv3x3(cur, 2 * B, p.h[k + 1], p.w[k + 1], layer, packed, prec, dst, st))) {
        return rc;
      }
      cur = dst;
      if (cs.last) {
        double *pk = part + p.sd.part_off[k + 1];
        if (k < 4) {
          if ((rc = pool_stats(cur, B, p.h[k + 1], p.w[k + 1], cs.cout, prec, bufA, pk, st))) return rc;
          cur = bufA;
        } else if ((rc = stats_nhwc(cur, B, p.h[k + 1] * p.w[k + 1], cs.cout, prec, pk, st))) {
          return rc;
        }
      }
    }
  }
  // ---- per-channel scalars, stage-0 images as NHWC4, entropies, channel weights ----
  {
    dim3 grid(cdiv(ctot, 256), B);
    TimedLaunch t(NQA_K_ADISTS, st);
    adists_prep_kernel<<<grid, 256, 0, st>>>(part, p.sd, q, B);
    if ((rc = check_launch("adists_prep"))) return rc;
  }
  float *img4x = reinterpret_cast<float *>(base + p.img4x), *img4y = reinterpret_cast<float *>(base + p.img4y);
  {
    dim3 grid(cdiv(H * W, 256), B);
    TimedLaunch t(NQA_K_ADISTS, st);
    nchw3_to_nhwc4_kernel<<<grid, 256, 0, st>>>(x, img4x, H * W);
    nchw3_to_nhwc4_kernel<<<grid, 256, 0, st>>>(y, img4y, H * W);
    if ((rc = check_launch("nchw3_to_nhwc4"))) return rc;
  }
  const size_t qst = (size_t)B * ctot;
  // stage 0's q/wgt rows are indexed with channel < 3; the NHWC4 kernels read index 3 too, which
  // is channel 0 of stage 1 in the concatenated vector -- harmless: its feature value is 0 (entropy
  // term 0) and the window kernel masks c >= creal.
  if ((rc = launch_entropy<PrecF32>(img4x, B, H * W, 4, p.ent_ppb[0], q + 0 * qst + p.sd.coff[0],
                                    q + 2 * qst + p.sd.coff[0], ctot, ent + p.ed.part_off[0], st)))
    return rc;
  for (int k = 1; k < 6; ++k) {
    const int hw = p.h[k] * p.w[k];
    const float *invx = q + 0 * qst + p.sd.coff[k], *sumx = q + 2 * qst + p.sd.coff[k];
    double *ep = ent + p.ed.part_off[k];
    switch (storage_prec(prec)) {
      case NQA_PREC_F32: rc = launch_entropy<PrecF32>(taps[k - 1], B, hw, p.c[k], p.ent_ppb[k], invx, sumx, ctot, ep, st); break;
      case NQA_PREC_BF16: rc = launch_entropy<PrecBF16>(taps[k - 1], B, hw, p.c[k], p.ent_ppb[k], invx, sumx, ctot, ep, st); break;
      default: rc = launch_entropy<PrecF16>(taps[k - 1], B, hw, p.c[k], p.ent_ppb[k], invx, sumx, ctot, ep, st); break;
    }
    if (rc) return rc;
  }
  {
    // hsum reuses the q[2] (sum_x) rows: they are dead once the entropy kernels have run
    float *hsum = q + 2 * qst;
    TimedLaunch t(NQA_K_ADISTS, st);
    adists_entropy_fold_kernel<<<dim3(cdiv(ctot, 4), B), 256, 0, st>>>(ent, p.ed, hsum);
    adists_weights_kernel<<<B, 256, 0, st>>>(hsum, p.ed, wgt);
    if ((rc = check_launch("adists_weights"))) return rc;
  }
  // ---- heavy pass: gamma / TW / SW maps per stage ----
  static const Gauss gauss = make_gauss();
  for (int i = 0; i < kWin; ++i)
    if (gauss.g[i] != kG[i]) {
      set_error("adists_forward: this host's exp() gives a different Gaussian window than the kernels' constants "
                "(tap %d: %a vs %a)", i, (double)gauss.g[i], (double)kG[i]);
      return NQA_E_LAUNCH;
    }
  for (int k = 0; k < 6; ++k) {
    float *gamma = reinterpret_cast<float *>(base + p.maps[k][0]);
    float *tw = reinterpret_cast<float *>(base + p.maps[k][1]);
    float *sw = reinterpret_cast<float *>(base + p.maps[k][2]);
    if (!p.windowed[k]) {
      TimedLaunch t(NQA_K_ADISTS, st);
      adists_global_kernel<<<B, 256, 0, st>>>(q, B, ctot, p.sd.coff[k], p.c[k], wgt, gamma, tw, sw);
      if ((rc = check_launch("adists_global"))) return rc;
      continue;
    }
    if (k == 0) {
      rc = launch_window_planar(x, y, B, H, W, q, ctot, 0, wgt, gauss, gamma, tw, sw, st);
    } else {
      const char *tx = static_cast<const char *>(taps[k - 1]);
      const char *ty = tx + (size_t)B * p.h[k] * p.w[k] * p.c[k] * esz;
      switch (storage_prec(prec)) {
        case NQA_PREC_F32: rc = launch_window_lanes<PrecF32>(tx, ty, B, p.h[k], p.w[k], p.c[k], q, ctot, p.sd.coff[k], wgt, gauss, gamma, tw, sw, st); break;
        case NQA_PREC_BF16: rc = launch_window_lanes<PrecBF16>(tx, ty, B, p.h[k], p.w[k], p.c[k], q, ctot, p.sd.coff[k], wgt, gauss, gamma, tw, sw, st); break;
        default: rc = launch_window_lanes<PrecF16>(tx, ty, B, p.h[k], p.w[k], p.c[k], q, ctot, p.sd.coff[k], wgt, gauss, gamma, tw, sw, st); break;
      }
    }
    if (rc) return rc;
  }
  // ---- probability chain, coarse to fine ----
  {
    TimedLaunch t(NQA_K_ADISTS, st);
    chain_init_kernel<<<cdiv(6 * B, 256), 256, 0, st>>>(acc, 6 * B);
    // initial ps_prod = ones (ADISTS.py:75): a 1x1 map of 1 upsamples to the same constant
    float *ones = reinterpret_cast<float *>(base + p.bufB);  // the ping-pong buffers are free now
    ones_kernel<<<cdiv(B, 256), 256, 0, st>>>(ones, B);
    const float *prev = ones;
    int hp = 1, wp = 1;
    for (int k = 5; k >= 0; --k) {
      const float *gamma = reinterpret_cast<const float *>(base + p.maps[k][0]);
      const float *tw = reinterpret_cast<const float *>(base + p.maps[k][1]);
      const float *sw = reinterpret_cast<const float *>(base + p.maps[k][2]);
      float *psprod = reinterpret_cast<float *>(base + p.maps[k][3]);
      ChainAcc *a = acc + (size_t)k * B;
      if (p.windowed[k]) {
        const int n = p.mh[k] * p.mw[k];
        dim3 grid(min(cdiv(n, 256), kChainBlocks), B);
        double *cp = reinterpret_cast<double *>(base + p.chain_part);
        chain_moments_kernel<<<grid, 256, 0, st>>>(gamma, n, cp);
        chain_fold_kernel<<<B, 256, 0, st>>>(cp, grid.x, 2, 0, a);
        chain_psminmax_kernel<<<grid, 256, 0, st>>>(gamma, n, a);
        chain_ppminmax_kernel<<<grid, 256, 0, st>>>(gamma, p.mh[k], p.mw[k], prev, hp, wp, a);
        chain_final_kernel<<<grid, 256, 0, st>>>(gamma, p.mh[k], p.mw[k], prev, hp, wp, tw, sw, psprod, a, cp);
        chain_fold_kernel<<<B, 256, 0, st>>>(cp, grid.x, 1, 1, a);
      } else {
        chain_global_kernel<<<cdiv(B, 256), 256, 0, st>>>(gamma, prev, hp * wp, tw, sw, psprod, a, B);
      }
      prev = psprod;
      hp = p.mh[k];
      wp = p.mw[k];
    }
    DDesc dd;
    for (int k = 0; k < 6; ++k) dd.n[k] = p.mh[k] * p.mw[k];
    adists_d_kernel<<<cdiv(B, 256), 256, 0, st>>>(acc, dd, B, d_out);
    if ((rc = check_launch("adists_chain"))) return rc;
  }
  if (map_out) {
    MapDesc md;
    for (int k = 0; k < 6; ++k) {
      md.tw[k] = reinterpret_cast<const float *>(base + p.maps[k][1]);
      md.sw[k] = reinterpret_cast<const float *>(base + p.maps[k][2]);
      md.ps[k] = reinterpret_cast<const float *>(base + p.maps[k][3]);
      md.mh[k] = p.mh[k];
      md.mw[k] = p.mw[k];
    }
    TimedLaunch t(NQA_K_ADISTS, st);
    adists_map_kernel<<<dim3(cdiv(H * W, 256), B), 256, 0, st>>>(md, H, W, map_out);
    if ((rc = check_launch("adists_map"))) return rc;
  }
  return NQA_OK;
}

extern "C" {

int nqa_adists_forward(const float *x, const float *y, int B, int H, int W, const void *packed, int prec, void *ws,
                       size_t ws_bytes, float *d_out, void *stream) {
  return adists_run(x, y, B, H, W, packed, prec, ws, ws_bytes, d_out, nullptr, stream);
}

int nqa_adists_forward_map(const float *x, const float *y, int B, int H, int W, const void *packed, int prec,
                           void *ws, size_t ws_bytes, float *d_out, float *map_out, void *stream) {
  if (!map_out) {
    set_error("adists_forward_map: null map pointer");
    return NQA_E_ARG;
  }
  return adists_run(x, y, B, H, W, packed, prec, ws, ws_bytes, d_out, map_out, stream);
}

}  // extern "C"
