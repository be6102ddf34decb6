// Stage 1 of the DISTS path in ONE kernel (gfx950, f16): input normalisation, conv1_1 + ReLU, conv1_2 + ReLU, the
// L2-pool in front of stage 2 and the five statistics sums of tap relu1_2 (DISTS_pt.py:92-94, :22-25, :130-142).
// Neither relu1_1 nor the 64-channel full-resolution tap relu1_2 (4.25 GB per B=8 1080p step, written by
// conv1_regw_kernel and read back by pool_stats_kernel until round 3) ever leaves the chip.
//
// Structure = nqa_conv_pool.hip's (x | y side by side in the MFMA column groups, register statistics, lane-local vertical
// and DPP horizontal pool, strips walked top to bottom with the row above carried in registers, column seams through
// fp32 partials, the epilogue of pass p issued inside the k loop of pass p + 1) on conv1_regw_kernel's data path (the
// halo image of relu1_1 is COMPUTED: raw pixels by LDS-DMA into a staging area two units ahead, normalised into a raw
// patch one unit ahead, conv1_1 on the matrix cores straight into the halo image conv1_2 reads):
//   * one wave per SIMD (4 waves, the whole register file): a wave = 32 of the 64 output channels (2 x 18 weight
//     fragments, 144 registers; conv1_1's 8 fragments and biases in registers too) x one 16-column half of a 32-column
//     unit x 4 rows x {x, y}.  A pixel fragment read from LDS feeds two MFMAs (the two 16-channel tiles), and the two
//     waves of a half read the same fragments: 0.5 LDS reads per MFMA, as in every register-weights kernel here;
//   * conv1_1 of unit u + 1 has no partner wave to hide under: it is cut into slices like the epilogue (fragment
//     reads -> 8 MFMAs -> ReLU / convert / zero outside the image / 4 LDS writes per 16 halo pixels) and rides in the
//     k loops of unit u, two groups of 16 halo pixels per pass;
//   * halo image: [image][6 x 34 pixels] records of 160 bytes (64 channels as two 32-channel chunks + 32 B pad): a
//     ds_read_b128 of 16 consecutive pixels is bank-conflict-free at that pitch, every tap an immediate offset.
#include <atomic>
#include <type_traits>

#include "nqa_common.h"

namespace nqa {

typedef __attribute__((address_space(3))) void lds_void_q;

__device__ static inline float dpp1_row_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ static inline float dpp1_row_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

struct S1Geom {
  static constexpr int TH = 4, TW = 32, HWD = TW + 2, NQI = (TH + 2) * HWD;  // 204 halo pixels per image
  static constexpr int NGI = (NQI + 15) / 16, NGRP = 2 * NGI;                 // 13 groups of 16 per image, 26 per unit
  static constexpr int PITCH = 160, IMG_BYTES = NQI * PITCH, SLOT = 2 * IMG_BYTES;  // 65 280 per halo slot
  static constexpr int RAWP = 40, RAW_ROWS = TH + 5, RAW_IMG = RAW_ROWS * RAWP * 8, RAW_BYTES = 2 * RAW_IMG;  // (+1 row read with zero weights)
  static constexpr int RAW_PX = (TH + 4) * (TW + 4);                         // 288 raw pixels per image
  static constexpr int STG_SLOTS = 12 * 64;                                  // staging: 3 rounds x 4 waves x 64 lanes per plane
  // DUMP: where conv1_1 groups that do not exist write (lane * 8 + up to 104 bytes of tile offset)
  static constexpr int RAW_OFF = 2 * SLOT, STG_OFF = RAW_OFF + 2 * RAW_BYTES, DUMP_OFF = STG_OFF + 3 * STG_SLOTS * 4;
  // XB: a pass's relu1_2 tile on its way from the wave that computed it to the wave that pools and sums it lives in the
  // 32-byte PADS of the halo records (never touched by conv1_1's writes or conv1_2's reads): lane (wave w, l) owns the pad
  // of record 64 w + l of halo slot `pass & 1` -- two buffers at no LDS cost, so ONE barrier per pass hands a tile over
  static constexpr int XB_PAD = 128;
  static constexpr int LDS = DUMP_OFF + 1024;
  static_assert(LDS <= 163840, "LDS budget");
};

// Eight waves, two per SIMD, in TWO ROLES (round 4, second form).  The first form ran everything in four waves, one per
// SIMD, with the epilogue and conv1_1 cut into slices between the MFMAs: correct, but a single wave issues one instruction
// per ~4 cycles whatever its kind, and at five vector / LDS / scalar instructions per MFMA the matrix pipe sat idle 63 %
// of the time (rocprofv3: profiles/r04_pmc_stage1_fused_vs_unfused.txt).  Now
//   * waves 0..3 (`conv`): conv1_2 only -- 72 MFMAs and 36 fragment reads per pass, the pass's 16 accumulators per lane
//     through ReLU to packed halves and into the exchange buffer XB;
//   * waves 4..7 (`tail`, wave w + 4 shares a SIMD with wave w): everything else for the SAME (channel group, half-strip)
//     -- the previous pass's tile from XB through the statistics, the pool and the stores, conv1_1 of the next unit into
//     the other halo image, the raw pixels;
// so the SIMD's arbiter interleaves one wave's MFMAs with the other's vector instructions by itself.  ONE barrier per
// pass hands a tile over: XB is double-buffered (in the pads of the halo records).
template <bool RAGGED>
__global__ __launch_bounds__(512) void conv1_pool_kernel(const float *__restrict__ x, const float *__restrict__ y, int B,
                                                         const char *__restrict__ w1m, const float *__restrict__ bias1,
                                                         const char *__restrict__ wreg, const float *__restrict__ bias2,
                                                         char *__restrict__ pooled, float *__restrict__ seam,
                                                         double *__restrict__ part, int H, int W, int spairs, int rows,
                                                         int total_units, int part_nblk) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef S1Geom G;
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  constexpr int GPP = 2, NKS = 18, NI = 2, NP = 8, COUT = 64;
  constexpr int NST = 8;  // stores per tail wave and pooled row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool tail = wave >= 4;           // (wave-uniform)
  const int w4 = wave & 3;
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = w4 & 1, hs = w4 >> 1;   // channel group (32 channels), half-strip (16 columns)
  const int HW = H * W;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const int strips = (W + 15) >> 4;      // 16-column strips (the seam planes' index)

  // ---- this block's run of units, in [pair][strip pair][row] order (see nqa_conv_pool.hip) ----
  const int nblk = gridDim.x;
  int run;
  {
    const int qq = nblk >> 3, rr = nblk & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    run = nblk < 8 ? (int)blockIdx.x : (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
  }
  const int u_lo = (int)((long)total_units * run / nblk), u_hi = (int)((long)total_units * (run + 1) / nblk);
  if (u_lo >= u_hi) return;  // (block-uniform)
  const bool warm0 = (u_lo % rows) != 0;  // the run starts inside a strip: one warm-up unit (the tile above), outputs dropped
  const int nsteps = (u_hi - u_lo) + (warm0 ? 1 : 0);
  // unit coordinates of steps s, s + 1, s + 2 (pair, strip pair, row), advanced by one unit per step: the divisions
  // happen once here (a scalar division is ~50 instructions)
  struct UC {
    int n, sp, ty;
  };
  UC uc[3], ucp;  // ucp: the previous step's (whose last row the tail completes first)
  {
    int u = u_lo - (warm0 ? 1 : 0);
    uc[0].ty = u % rows;
    u /= rows;
    uc[0].sp = u % spairs;
    uc[0].n = u / spairs;
  }
  auto next_uc = [&](const UC &a) {
    UC b = a;
    if (++b.ty == rows) {
      b.ty = 0;
      if (++b.sp == spairs) {
        b.sp = 0;
        ++b.n;
      }
    }
    return b;
  };
  uc[1] = next_uc(uc[0]);
  uc[2] = next_uc(uc[1]);
  ucp = uc[0];
  char *const xb = smem + (w4 * 64 + lane) * G::PITCH + G::XB_PAD;  // this lane's 32 bytes of exchange buffer 0 (+ SLOT: buffer 1)

  if (!tail) {
    // =============================== conv waves: conv1_2, nothing else ===============================
    u32x4 wf[2][NKS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cg * 2 + i) * NKS + ks) * 64 + lane) * 16);
    f32x4 bia[NI];  // the bias is the accumulators' initial value
#pragma unroll
    for (int i = 0; i < NI; ++i) bia[i] = *reinterpret_cast<const f32x4 *>(bias2 + cg * 32 + i * 16 + 4 * c4);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): retired with a wait the compiler can see (nqa_conv.hip, 4.2a)
    asm volatile("s_barrier" ::: "memory");  // P1: raw patches zeroed
    asm volatile("s_barrier" ::: "memory");  // P2: raw patches of steps 0, 1 committed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // P3: halo image of step 0 complete
    for (int step = 0; step < nsteps; ++step) {
      const int slot_off = (step & 1) * G::SLOT;
#pragma unroll 1
      for (int pass = 0; pass < 4; ++pass) {
        int rb = slot_off + (pass * G::HWD + hs * 16 + l15) * G::PITCH + (c4 << 4);
        asm volatile("" : "+v"(rb));
        f32x4 acc[2][GPP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < GPP; ++g) acc[i][g] = bia[i];
        u32x4 bf[2][GPP];
        auto load_b = [&](int ks, u32x4(&b)[GPP]) {
          const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
          for (int g = 0; g < GPP; ++g)
            b[g] = *reinterpret_cast<const u32x4 *>(smem + rb + (g * G::IMG_BYTES + (ky * G::HWD + kx) * G::PITCH + cc * 64));
        };
        load_b(0, bf[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks + 1 < NKS) load_b(ks + 1, bf[(ks + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#ifndef NQA_S1_NO_MFMA  // (timing-only ablations, tools/gpu_s1_ablate.sh: results are wrong on purpose)
#pragma unroll
          for (int g = 0; g < GPP; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                                 __builtin_bit_cast(f16x8, bf[ks & 1][g]), acc[i][g], 0, 0, 0);
#else
          if (ks == 0)
#pragma unroll
            for (int g = 0; g < GPP; ++g) asm volatile("" ::"v"(bf[0][g]), "v"(bf[1][g]));
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        // ReLU on packed halves; piece i = (tile i: image 0 channels e, image 1 channels e)
        const h2 z = {(_Float16)0.f, (_Float16)0.f};
        u32x4 piece[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g = 0; g < GPP; ++g) {
            const h2 lo = __builtin_elementwise_max(__builtin_convertvector((f2){acc[i][g][0], acc[i][g][1]}, h2), z);
            const h2 hi = __builtin_elementwise_max(__builtin_convertvector((f2){acc[i][g][2], acc[i][g][3]}, h2), z);
            piece[i][2 * g] = __builtin_bit_cast(unsigned, lo);
            piece[i][2 * g + 1] = __builtin_bit_cast(unsigned, hi);
          }
        // buffer pass & 1: its previous content (the tile of pass - 2) was read by the tail waves before the last barrier
        *reinterpret_cast<u32x4 *>(xb + (pass & 1) * G::SLOT) = piece[0];
        *reinterpret_cast<u32x4 *>(xb + (pass & 1) * G::SLOT + 16) = piece[1];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // this pass's tile is in XB
      }
    }
    return;
  }

  // =============================== tail waves ===============================
  // conv1_1's 8 fragments and its bias (the accumulators' initial value)
  u32x4 w1f[4][2];
  f32x4 b1v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int m = 0; m < 2; ++m) w1f[i][m] = *reinterpret_cast<const u32x4 *>(w1m + ((i * 2 + m) * 64 + lane) * 16);
    b1v[i] = *reinterpret_cast<const f32x4 *>(bias1 + 16 * i + 4 * c4);
  }
  const int ttid = tid - 256;  // 0..255 among the tail waves
  // zero both raw patches (their borders and the spare row are read with zero weights or as padding)
  for (int i = ttid; i < 2 * G::RAW_BYTES / 8; i += 256) reinterpret_cast<u32x2 *>(smem + G::RAW_OFF)[i] = (u32x2){0u, 0u};

  // ---- raw pixels: 2 images x 8 x 36 per unit; chunk c = round * 4 + wave holds 64 pixels of ONE image (5 chunks per
  // image, 288 of their 320 slots live), so a DMA instruction has one image = one buffer resource ----
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  const float isd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
  const unsigned kOOB = 0x80000000u;
  int r_img[3], r_row[3], r_col[3];
  bool r_live[3], r_ok[3] = {false, false, false};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int c = r * 4 + w4, p = (c % 5) * 64 + lane;
    r_img[r] = c / 5;
    r_live[r] = c < 10 && p < G::RAW_PX;
    r_row[r] = p / (G::TW + 4);
    r_col[r] = p - r_row[r] * (G::TW + 4);  // (once per kernel)
  }
  auto raw_fetch = [&](const UC &c, bool real) {  // (every tail wave issues its 9 pieces whatever: the counted waits rely on it)
    const int n = real ? c.n : 0;
    const int x0 = c.sp * G::TW, y0 = c.ty * G::TH;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int gy = y0 - 2 + r_row[r], gx = x0 - 2 + r_col[r];
      r_ok[r] = real && r_live[r] && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float *img = (r_img[r] ? y : x) + (size_t)n * 3 * HW;  // (wave-uniform)
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, 3u * (unsigned)HW * 4u, 0x00020000);
      const unsigned off = r_ok[r] ? (unsigned)((gy * W + gx) * 4) : kOOB;
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (lds_void_q *)(smem + G::STG_OFF + (c3 * G::STG_SLOTS + (r * 4 + w4) * 64) * 4), 4, off, c3 * HW * 4, 0, 0);
    }
  };
  auto raw_commit = [&](int buf) {  // staging -> normalised f16 raw patch `buf` (the caller has waited for the DMA)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (r_live[r]) {
        h4 v;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          const float raw = *reinterpret_cast<const float *>(smem + G::STG_OFF + (c3 * G::STG_SLOTS + (r * 4 + w4) * 64 + lane) * 4);
          // (x - mean) / std as a multiplication by 1 / std with one residual correction: 4 instructions instead of ~10
          const float d = raw - mean[c3], q0 = d * isd[c3], q1 = fmaf(fmaf(-q0, sd[c3], d), isd[c3], q0);
          v[c3] = (_Float16)(r_ok[r] ? q1 : 0.f);
        }
        v[3] = (_Float16)0.f;
        *reinterpret_cast<h4 *>(smem + G::RAW_OFF + buf * G::RAW_BYTES + r_img[r] * G::RAW_IMG +
                                (r_row[r] * G::RAWP + r_col[r]) * 8) = v;
      }
    }
  };

  // ---- conv1_1 of one group of 16 halo pixels -> the halo image of the unit `c11_*` describe ----
  // group gi = w4 + 4 j (j = 0..6; 26 groups): image gi / 13, halo pixels (gi % 13) * 16 + l15 of its 204
  int c11_y0 = 0, c11_x0 = 0, c11_par = 0;
  bool c11_real = false;
  auto c11_set = [&](const UC &c, bool real, int step) {
    c11_real = real;
    c11_y0 = c.ty * G::TH;
    c11_x0 = c.sp * G::TW;
    c11_par = step & 1;
  };
  // NG groups at once (j0, j0 + 1, ...): their read -> 8 MFMAs -> convert -> write chains interleave, which is all the
  // latency hiding a wave gets whose SIMD partner is busy with conv1_2
  // what a group's lane needs is fixed for the kernel's life but for the unit's origin and parity: per group j the byte
  // offset of its raw fragment (parity 0), of its halo record (parity 0; negative: the group / pixel does not exist),
  // and its halo row / column -- 21 registers instead of ~25 address instructions per group and unit
  int t_raw[7], t_dst[7], t_yx[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int gi = w4 + 4 * j;
    const int img = gi >= G::NGI ? 1 : 0, q = (gi - img * G::NGI) * 16 + l15, qc = q < G::NQI ? q : G::NQI - 1;
    const int hy = qc / G::HWD, hx = qc - hy * G::HWD;
    t_yx[j] = (hy << 8) | hx;
    t_raw[j] = G::RAW_OFF + img * G::RAW_IMG + (c4 & 1) * 16 + ((hy + (c4 >> 1)) * G::RAWP + hx) * 8;
    t_dst[j] = (gi < G::NGRP && q < G::NQI) ? img * G::IMG_BYTES + q * G::PITCH + (c4 << 3) : -1;
  }
  auto c11_groups = [&](int j0, auto ng_tag) {  // (j0 is a compile-time constant at every call)
    constexpr int NG = decltype(ng_tag)::value;
    int dst[NG];
    unsigned mask[NG];
    u32x4 bfr[NG][2];
    f32x4 a1[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int j = j0 + u;
      const int gy = c11_y0 - 1 + (t_yx[j] >> 8), gx = c11_x0 - 1 + (t_yx[j] & 255);
      const bool inside = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
      mask[u] = inside ? 0xFFFFFFFFu : 0u;
      dst[u] = (c11_real & (t_dst[j] >= 0)) ? c11_par * G::SLOT + t_dst[j] : G::DUMP_OFF + lane * 8;
      const unsigned rawb = (unsigned)(t_raw[j] + c11_par * G::RAW_BYTES);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const char *rp = smem + rawb + m * (2 * G::RAWP * 8);
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(rp), hi = *reinterpret_cast<const u32x2 *>(rp + 8);
        bfr[u][m] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)  // kernel rows 0-1 (the accumulator starts at the bias) ...
#pragma unroll
      for (int u = 0; u < NG; ++u)
        a1[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w1f[i][0]), __builtin_bit_cast(f16x8, bfr[u][0]), b1v[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)  // ... then row 2 (+ zeros)
#pragma unroll
      for (int u = 0; u < NG; ++u)
        a1[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w1f[i][1]), __builtin_bit_cast(f16x8, bfr[u][1]), a1[u][i], 0, 0, 0);
    // ReLU and the zero outside the image (conv1_2's padding) on packed halves
    const h2 z = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // channels 16 i + 4 c4 ..: chunk i >> 1, quarter 2 (i & 1) + (c4 >> 1), half (c4 & 1)
        const h2 lo = __builtin_elementwise_max(__builtin_convertvector((f2){a1[u][i][0], a1[u][i][1]}, h2), z);
        const h2 hi = __builtin_elementwise_max(__builtin_convertvector((f2){a1[u][i][2], a1[u][i][3]}, h2), z);
        const u32x2 vv = {__builtin_bit_cast(unsigned, lo) & mask[u], __builtin_bit_cast(unsigned, hi) & mask[u]};
        *reinterpret_cast<u32x2 *>(smem + dst[u] + (i >> 1) * 64 + (i & 1) * 32) = vv;
      }
  };
  typedef std::integral_constant<int, 2> Two;
  typedef std::integral_constant<int, 1> One;

  // ---- per-lane state: vertical pool sums, shifted statistics (nqa_conv_pool.hip) ----
  float U[NI][GPP][4];
  float piv[NP], s1x[NP], s1y[NP], s2x[NP], s2y[NP], sxy[NP], n_lane = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < GPP; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) U[i][g][e] = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) piv[p] = s1x[p] = s1y[p] = s2x[p] = s2y[p] = sxy[p] = 0.f;
  const bool is15 = l15 == 15, is0 = l15 == 0, even_in = (l15 & 1) == 0 && l15 != 0;
  const size_t nimg = 2 * (size_t)B;
  const unsigned pooled_bytes = (unsigned)(nimg * Ho * Wo * COUT * 2);            // (host-checked < 2^31)
  const unsigned seam_plane = (unsigned)(nimg * strips * Ho * COUT);              // elements; two planes: own | left
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(pooled, 0, pooled_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(seam, 0, 2u * seam_plane * 4u, 0x00020000);
  const unsigned y_pool_off = (unsigned)((size_t)B * Ho * Wo * COUT * 2);
  const unsigned y_seam_off = (unsigned)((size_t)B * strips * Ho * COUT);
  const int ch_lane0 = cg * 32 + 4 * c4;

  // the tile of one pass out of XB: v[i][g][e] as floats (the tap's f16 values)
  float v[NI][GPP][4];
  auto take_tile = [&](int buf) {
    // (an LDS-address-space VOLATILE read: the pieces are written by other waves between two barriers.  Through the generic
    // pointer hipcc re-read only the first dword of each piece inside the unit loop and carried the other three over from
    // the previous pass; a volatile generic access turns into flat loads, which the counted vmcnt waits cannot live with.)
    typedef __attribute__((address_space(3))) const volatile u32x4 lds_cv4;
    u32x4 pc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) pc[i] = *reinterpret_cast<lds_cv4 *>((lds_void_q *)(xb + buf * G::SLOT + i * 16));
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const h2 lo = __builtin_bit_cast(h2, (uint32_t)pc[i][2 * g]), hi = __builtin_bit_cast(h2, (uint32_t)pc[i][2 * g + 1]);
        v[i][g][0] = (float)lo[0];
        v[i][g][1] = (float)lo[1];
        v[i][g][2] = (float)hi[0];
        v[i][g][3] = (float)hi[1];
      }
  };
  auto stats_tile = [&](float vmask) {
    n_lane += RAGGED ? vmask : 1.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = p >> 2, e = p & 3;
      float dx = v[i][0][e] - piv[p], dy = v[i][1][e] - piv[p];
      if constexpr (RAGGED) {
        dx *= vmask;
        dy *= vmask;
      }
      s1x[p] += dx;
      s1y[p] += dy;
      s2x[p] = fmaf(dx, dx, s2x[p]);
      s2y[p] = fmaf(dy, dy, s2y[p]);
      sxy[p] = fmaf(dx, dy, sxy[p]);
    }
  };
  // even rows (0, 2): U = (row above) + 2 s
  auto pool_even = [&](float vmask) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m = RAGGED ? v[i][g][e] * vmask : v[i][g][e];
          U[i][g][e] = fmaf(2.f * m, m, U[i][g][e]);
        }
  };
  // odd rows (1, 3): U += s completes pooled row (row - 1) / 2: horizontal taps by DPP, root, stores; U = s carries on
  auto pool_odd = [&](float vmask, unsigned po_lane, unsigned so_lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      h4 outh[GPP];
      f32x4 seamv[GPP];
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m = RAGGED ? v[i][g][e] * vmask : v[i][g][e];
          const float sq = m * m;
          const float u = U[i][g][e] + sq;
          const float a = u + dpp1_row_shr1(u);
          const float pv = a + dpp1_row_shl1(a);
          seamv[g][e] = is15 ? u : pv;
          outh[g][e] = (_Float16)__builtin_amdgcn_sqrtf(fmaf(pv, 0.0625f, 1e-12f));
          U[i][g][e] = sq;
        }
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        // (an out-of-range offset stays out of range under the small additions: every tail wave issues the same stores)
        const unsigned po = po_lane + (g ? y_pool_off : 0u) + (unsigned)(i * 16) * 2u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, outh[g]), prsrc, po, 0, 0);
        const unsigned so = so_lane + (g ? y_seam_off * 4u : 0u) + (unsigned)(i * 16) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, seamv[g]), srsrc, so, 0, 0);
      }
    }
  };
  // where pooled row oy = y0 / 2 + j of unit `c` goes, for this lane (kOOB: nowhere)
  // (lane constants: this lane's part of the pooled / seam byte offsets; kOOB where the lane never stores)
  const unsigned lane_po = even_in ? (unsigned)(((l15 >> 1) * COUT + ch_lane0) * 2) : kOOB;
  const unsigned lane_so0 = is0 ? (unsigned)ch_lane0 * 4u : kOOB;
  const unsigned lane_so15 = is15 ? (seam_plane + (unsigned)(Ho * COUT) + (unsigned)ch_lane0) * 4u : kOOB;
  auto pool_offsets = [&](const UC &c, int j, bool emit, unsigned &po_lane, unsigned &so_lane) {
    const int s16 = c.sp * 2 + hs, x0 = s16 * 16, oy = ((c.ty * G::TH) >> 1) + j, ox0 = x0 >> 1;
    const bool live = emit && oy < Ho && x0 < W;  // (block-uniform)
    const unsigned pool_row = (((unsigned)c.n * (unsigned)Ho + (unsigned)oy) * (unsigned)Wo + (unsigned)ox0) * (COUT * 2u);
    const unsigned seam_row = (((unsigned)c.n * (unsigned)strips + (unsigned)s16) * (unsigned)Ho + (unsigned)oy) * (COUT * 4u);
    const bool right = s16 + 1 < strips;
    // (kOOB + a row offset below 2^31 stays out of range)
    po_lane = live ? lane_po + pool_row : kOOB;
    if constexpr (RAGGED) po_lane = (ox0 + (l15 >> 1) < Wo) ? po_lane : kOOB;
    so_lane = live ? (right ? (is15 ? lane_so15 : lane_so0) : lane_so0) + seam_row : kOOB;
  };
  auto row_mask = [&](const UC &c, int pass) -> float {
    if constexpr (!RAGGED) return 1.f;
    const int x0 = (c.sp * 2 + hs) * 16, y0 = c.ty * G::TH;
    return (y0 + pass < H && x0 + l15 < W) ? 1.f : 0.f;
  };
  auto reset_stats = [&]() {  // (v holds row 0 of the pair's first tile: the pivots)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      piv[p] = v[p >> 2][0][p & 3];
      s1x[p] = s1y[p] = s2x[p] = s2y[p] = sxy[p] = 0.f;
    }
    n_lane = 0.f;
  };
  auto flush_stats = [&](int n) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = p >> 2, e = p & 3;
      const double pv = piv[p], nn = n_lane, ax = s1x[p], ay = s1y[p];
      double r[5] = {ax + nn * pv, ay + nn * pv, (double)s2x[p] + 2.0 * pv * ax + nn * pv * pv,
                     (double)s2y[p] + 2.0 * pv * ay + nn * pv * pv, (double)sxy[p] + pv * ax + pv * ay + nn * pv * pv};
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) r[s5] += __shfl_xor(r[s5], m, 16);
      }
      // the two half-strip waves of a channel group hold sums of the same channels: rows 2 * block + hs of the partials
      if (l15 == 0) {
        const int c = ch_lane0 + i * 16 + e;
        double *dst = part + (((size_t)n * part_nblk + 2 * blockIdx.x + hs) * COUT + c) * 5;
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) dst[s5] = r[s5];
      }
    }
  };

  // ---- prologue: raw patches of steps 0 and 1, the halo image of step 0 ----
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // P1
  raw_fetch(uc[0], true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_commit(0);
  raw_fetch(uc[1], 1 < nsteps);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_commit(1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // P2
  c11_set(uc[0], true, 0);
  c11_groups(0, Two());
  c11_groups(2, Two());
  c11_groups(4, Two());
  c11_groups(6, One());
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // P3

  int flush_n = -1;
  bool have_prev = false;  // XB holds a real tile (from the second pass on)
  for (int step = 0; step < nsteps; ++step) {
    const int n = uc[0].n;
    const bool warm = warm0 && step == 0;
    const bool fresh = step == (warm0 ? 1 : 0);
    const bool change = !fresh && !warm && n != flush_n;
    bool flushed = false;
    raw_fetch(uc[2], step + 2 < nsteps);  // lands while this unit is computed
    c11_set(uc[1], step + 1 < nsteps, step + 1);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {  // (unrolled: static group indices, row parities and branches)
      // the tile of the PREVIOUS pass (pass 0: the previous step's row 3) is in XB since the last barrier
#ifdef NQA_S1_NO_EPI
      if (false) {
#else
      if (have_prev) {
#endif
        take_tile((pass + 1) & 1);
        const UC &tc = pass == 0 ? ucp : uc[0];
        const int tp = (pass + 3) & 3;                      // that tile's row in its unit
        const bool temit = pass == 0 ? !(warm0 && step == 1) : !warm;
        if (pass == 1) {
          // row 0 of this step: where a strip starts the carried row above is the zero padding; where a pair starts
          // (or the run does) the lane sums change hands
          if (uc[0].ty == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
              for (int g = 0; g < GPP; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) U[i][g][e] = 0.f;
          }
          if (fresh || change) {
            if (change) {
              flush_stats(flush_n);
              flushed = true;
            }
            if (!warm) {
              reset_stats();
              flush_n = n;
            }
          }
        }
        const float vm = row_mask(tc, tp);
        stats_tile(vm);
        if (tp & 1) {
          unsigned po_lane, so_lane;
          pool_offsets(tc, tp >> 1, temit, po_lane, so_lane);
          pool_odd(vm, po_lane, so_lane);
        } else {
          pool_even(vm);
        }
      } else if (pass == 1) {  // (a run's very first tile: nothing carried, nothing to hand over)
        take_tile(0);
        if (!warm) {
          reset_stats();
          flush_n = n;
        }
        stats_tile(row_mask(uc[0], 0));
        pool_even(row_mask(uc[0], 0));
      }
      if (pass >= 1) have_prev = true;
      // conv1_1 of the next unit: two groups per pass (one in the last)
#ifndef NQA_S1_NO_C11
      if (pass == 0) c11_groups(0, Two());
      if (pass == 1) c11_groups(2, Two());
      if (pass == 2) c11_groups(4, Two());
      if (pass == 3) c11_groups(6, One());
#endif
      if (pass == 3) {
        // the raw pixels of step + 2 were requested at the start of this unit; the only younger operations of this wave
        // are the unit's 2 * NST stores (and, rarely, a flush's): retire the pixels without waiting for the stores
        if (flushed)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (step == 0)  // (the run's first unit issued one pooled row's stores, not two)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");
        raw_commit(step & 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // this pass's LDS writes are out; the conv waves' new tile is in XB
    }
    ucp = uc[0];
    uc[0] = uc[1];
    uc[1] = uc[2];
    uc[2] = next_uc(uc[2]);
  }
  // ---- drain: the last unit's row 3, then this wave's last sums ----
  {
    take_tile(1);
    const float vm = row_mask(ucp, 3);
    stats_tile(vm);
    unsigned po_lane, so_lane;
    pool_offsets(ucp, 1, !(warm0 && nsteps == 1), po_lane, so_lane);
    pool_odd(vm, po_lane, so_lane);
  }
  if (flush_n >= 0) flush_stats(flush_n);
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int s1_num_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  dev &= 63;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (!n) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

static thread_local int g_fuse_stage1 = 1;  // nqa_set_conv_variant bit 7 (128): 0 = the unfused stage 1 + pool + statistics
void set_fuse_stage1(int on) { g_fuse_stage1 = on & 1; }

bool conv1_pool_fusable(int B, int H, int W, int blob_prec) {
  if (!g_fuse_stage1 || blob_prec != NQA_PREC_F16 || W < 16 || H < 4) return false;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, strips = cdiv(W, 16);
  if ((size_t)H * W * 3 * 4 >= (1ull << 31)) return false;
  if (2ull * B * Ho * Wo * 64 * 2 >= (1ull << 31)) return false;
  if (2ull * 2 * B * strips * Ho * 64 * 4 >= (1ull << 31)) return false;
  return true;
}

template <bool RAGGED>
static int launch_conv1_pool(const float *x, const float *y, int B, int H, int W, const char *packed, void *pooled, float *seam,
                             double *part, hipStream_t st) {
  typedef S1Geom G;
  static std::atomic<bool> attr_done_dev[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<bool> &attr_done = attr_done_dev[dev & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_pool_kernel<RAGGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS) != hipSuccess) {
      set_error("conv1_pool: cannot raise the dynamic LDS limit to %d bytes", G::LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = s1_num_cus();
  if (!cus) {
    set_error("conv1_pool: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int spairs = cdiv(W, G::TW), rows = cdiv(H, G::TH);
  const long units = (long)B * spairs * rows;
  int grid = (int)(units < cus ? units : cus);
  if (2 * grid > NQA_FUSED_PART_BLOCKS_S1) grid = NQA_FUSED_PART_BLOCKS_S1 / 2;  // (two partial rows per block: the half-strip waves)
  if (hipMemsetAsync(part, 0, (size_t)B * NQA_FUSED_PART_BLOCKS_S1 * 64 * 5 * sizeof(double), st) != hipSuccess) {
    set_error("conv1_pool: memset of the statistics partials failed");
    return NQA_E_LAUNCH;
  }
  const float *b1 = reinterpret_cast<const float *>(packed + layer_bias_offset(0, NQA_PREC_F16));
  const float *b2 = reinterpret_cast<const float *>(packed + layer_bias_offset(1, NQA_PREC_F16));
  {
    TimedLaunch t(NQA_K_CONV, st);
    conv1_pool_kernel<RAGGED><<<grid, 512, G::LDS, st>>>(x, y, B, packed + layer0_m16_offset(NQA_PREC_F16), b1,
                                                         packed + regw_offset(1, NQA_PREC_F16), b2, static_cast<char *>(pooled),
                                                         seam, part, H, W, spairs, rows, (int)units, NQA_FUSED_PART_BLOCKS_S1);
    const int rc = check_launch("conv1_pool");
    if (rc) return rc;
  }
  return pool_seam_finish(seam, pooled, 2 * B, cdiv(W, 16), (H + 1) / 2, (W + 1) / 2, 64, st);
}

// stage 1 of the B pairs (x, y: fp32 NCHW) -> pooled relu1_2 (2B images, NHWC f16, 64 channels) + statistics partials
int conv1_pool_stats_fused(const float *x, const float *y, int B, int H, int W, const void *packed, void *pooled, float *seam,
                           double *part, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  const bool ragged = (H % 4) != 0 || (W % 32) != 0;  // (a unit is 32 columns: its second half-strip may lie outside the image)
  return ragged ? launch_conv1_pool<true>(x, y, B, H, W, p, pooled, seam, part, st)
                : launch_conv1_pool<false>(x, y, B, H, W, p, pooled, seam, part, st);
}

}  // namespace nqa
