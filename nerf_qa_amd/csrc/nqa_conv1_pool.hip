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
  // DUMP: where the conv1_1 slices of groups that do not exist write (lane * 8 + up to 104 bytes of tile offset)
  static constexpr int RAW_OFF = 2 * SLOT, STG_OFF = RAW_OFF + 2 * RAW_BYTES, DUMP_OFF = STG_OFF + 3 * STG_SLOTS * 4;
  static constexpr int LDS = DUMP_OFF + 1024;
  static_assert(LDS <= 163840, "LDS budget");
};

// ROUND: the sums and the pool take relu1_2 ROUNDED to f16, exactly the values the unfused path stores and reads back (the
// tests' comparison form, nqa_set_conv_variant + 256); the shipped form skips the two conversions per value.
template <bool RAGGED, bool ROUND>
__global__ __launch_bounds__(256, 1) void conv1_pool_kernel(const float *__restrict__ x, const float *__restrict__ y, int B,
                                                            const char *__restrict__ w1m, const float *__restrict__ bias1,
                                                            const char *__restrict__ wreg, const float *__restrict__ bias2,
                                                            char *__restrict__ pooled, float *__restrict__ seam,
                                                            double *__restrict__ part, int H, int W, int spairs, int rows,
                                                            int total_units, int part_nblk) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  typedef S1Geom G;
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  constexpr int PF = 3, GPP = 2, NKS = 18, NI = 2, NP = 8, COUT = 64;
  constexpr int NST = 8;  // stores per wave and pooled row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = wave & 1, hs = wave >> 1;  // channel group (32 channels), half-strip (16 columns)
  const int HW = H * W;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const int strips = (W + 15) >> 4;  // 16-column strips (the seam planes' index)

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
  // happen once here, not in every pass (a scalar division is ~50 instructions of a wave that has the SIMD to itself)
  struct UC {
    int n, sp, ty;
  };
  UC uc[3];
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

  // ---- weights in registers: conv1_2 (2 tiles x 18 k-steps), conv1_1 (4 tiles x 2 MFMAs), both biases ----
  u32x4 wf[2][NKS];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cg * 2 + i) * NKS + ks) * 64 + lane) * 16);
  // the weight fragments belong in the ACCUMULATION half of the register file (an MFMA reads its A operand from there
  // directly): left to itself the allocator kept part of them in VGPRs and copied others in with v_accvgpr_read before use
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(wf[i][ks]));
  // conv1_1's 8 fragments and its bias (the accumulators' initial value): registers of the accumulation half too.  (Read
  // from LDS inside the slices, each fragment was waited for in place -- `ds_read; s_waitcnt lgkmcnt(0); v_mfma` eight
  // times per pass with no other wave on the SIMD to cover the latency: the first build of this kernel ran at half speed.)
  u32x4 w1f[4][2];
  f32x4 b1v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      w1f[i][m] = *reinterpret_cast<const u32x4 *>(w1m + ((i * 2 + m) * 64 + lane) * 16);
      asm volatile("" : "+a"(w1f[i][m]));
    }
    b1v[i] = *reinterpret_cast<const f32x4 *>(bias1 + 16 * i + 4 * c4);
    asm volatile("" : "+a"(b1v[i]));
  }
  // conv1_2's bias is the accumulators' initial value (no add in the epilogue, no zeroing move in front of a pass)
  f32x4 bia[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bia[i] = *reinterpret_cast<const f32x4 *>(bias2 + cg * 32 + i * 16 + 4 * c4);
    asm volatile("" : "+a"(bia[i]));
  }
  // zero both raw patches (their borders and the spare row are read with zero weights or as padding)
  for (int i = tid; i < 2 * G::RAW_BYTES / 8; i += 256) reinterpret_cast<u32x2 *>(smem + G::RAW_OFF)[i] = (u32x2){0u, 0u};

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
    const int c = r * 4 + wave, p = (c % 5) * 64 + lane;
    r_img[r] = c / 5;
    r_live[r] = c < 10 && p < G::RAW_PX;
    r_row[r] = p / (G::TW + 4);
    r_col[r] = p - r_row[r] * (G::TW + 4);  // (once per kernel)
  }
  auto raw_fetch = [&](const UC &c, bool real) {  // (every wave issues its 9 pieces whatever: the counted waits rely on it)
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
      for (int c = 0; c < 3; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (lds_void_q *)(smem + G::STG_OFF + (c * G::STG_SLOTS + (r * 4 + wave) * 64) * 4), 4, off, c * HW * 4, 0, 0);
    }
  };
  auto raw_commit = [&](int buf) {  // staging -> normalised f16 raw patch `buf` (the caller has waited for the DMA)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (r_live[r]) {
        h4 v;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float raw = *reinterpret_cast<const float *>(smem + G::STG_OFF + (c * G::STG_SLOTS + (r * 4 + wave) * 64 + lane) * 4);
          // (x - mean) / std as a multiplication by 1 / std with one residual correction: the quotient the division gives,
          // in 4 instructions instead of its ~10
          const float d = raw - mean[c], q0 = d * isd[c], q1 = fmaf(fmaf(-q0, sd[c], d), isd[c], q0);
          v[c] = (_Float16)(r_ok[r] ? q1 : 0.f);
        }
        v[3] = (_Float16)0.f;
        *reinterpret_cast<h4 *>(smem + G::RAW_OFF + buf * G::RAW_BYTES + r_img[r] * G::RAW_IMG +
                                (r_row[r] * G::RAWP + r_col[r]) * 8) = v;
      }
    }
  };

  // ---- conv1_1 of one group of 16 halo pixels, in three pieces (they ride in the k loops as slices) ----
  // group gi = wave + 4 j (j = 0..6; 26 groups): image gi / 13, halo pixels (gi % 13) * 16 + l15 of its 204
  f32x4 a1[4];
  u32x4 bfr[2];
  int c_q = 0, c_dst = G::DUMP_OFF;  // the group's halo pixel of this lane; LDS byte address of its record (or the dump)
  bool c_inside = false;
  unsigned c_mask = 0u;             // all ones where the group's pixel of this lane lies inside the image
  // the unit whose halo image is being produced (set per unit, outside the k loops: no branch inside them)
  int c11_y0 = 0, c11_x0 = 0, c11_par = 0;
  bool c11_real = false;
  auto c11_set = [&](const UC &c, bool real, int step) {
    c11_real = real;
    c11_y0 = c.ty * G::TH;
    c11_x0 = c.sp * G::TW;
    c11_par = step & 1;
  };
  auto c11_load = [&](int j) {
    // (full-rate integer arithmetic only: 24-bit multiplies and a multiply-shift division -- the 32-bit multiplies and
    // the 64-bit address arithmetic of the first build ran at a quarter of the rate, ~100 cycles per group)
    const int gi = wave + 4 * j;
    const int img = gi >= G::NGI ? 1 : 0, q = (gi - img * G::NGI) * 16 + l15, qc = q < G::NQI ? q : G::NQI - 1;
    const int hy = (int)(__umul24((unsigned)qc, 1928u) >> 16);  // qc / 34 for qc < 512
    const int hx = qc - (int)__umul24((unsigned)hy, (unsigned)G::HWD);
    const int gy = c11_y0 - 1 + hy, gx = c11_x0 - 1 + hx;
    c_inside = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
    c_mask = c_inside ? 0xFFFFFFFFu : 0u;
    const bool live = c11_real & (gi < G::NGRP) & (q < G::NQI);
    c_dst = live ? c11_par * G::SLOT + img * G::IMG_BYTES + (int)__umul24((unsigned)q, (unsigned)G::PITCH) + (c4 << 3)
                 : G::DUMP_OFF + lane * 8;
    c_q = q;
    const unsigned rawb = (unsigned)(G::RAW_OFF + c11_par * G::RAW_BYTES + img * G::RAW_IMG + (c4 & 1) * 16) +
                          (__umul24((unsigned)(hy + (c4 >> 1)), (unsigned)G::RAWP) + (unsigned)hx) * 8u;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const char *rp = smem + rawb + m * (2 * G::RAWP * 8);
      const u32x2 lo = *reinterpret_cast<const u32x2 *>(rp), hi = *reinterpret_cast<const u32x2 *>(rp + 8);
      bfr[m] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  // channel tile i, MFMA m: kernel rows 0-1 (m = 0, the accumulator starts at the bias), then row 2 + zeros (m = 1)
  auto c11_mma = [&](int i, int m) {
    if (m == 0)
      a1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w1f[i][0]), __builtin_bit_cast(f16x8, bfr[0]), b1v[i], 0, 0, 0);
    else
      a1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w1f[i][1]), __builtin_bit_cast(f16x8, bfr[1]), a1[i], 0, 0, 0);
  };
  // ReLU and the zero outside the image (conv1_2's padding) on PACKED halves: convert two at a time, v_pk_max_f16 against 0,
  // AND with the lane's all-ones / all-zeros mask -- 6 vector instructions per tile instead of 14
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  auto c11_store = [&](int i) {  // channels 16 i + 4 c4 ..: chunk i >> 1, quarter 2 (i & 1) + (c4 >> 1), half (c4 & 1)
    const h2 z = {(_Float16)0.f, (_Float16)0.f};
    const h2 lo = __builtin_elementwise_max(__builtin_convertvector((f2){a1[i][0], a1[i][1]}, h2), z);
    const h2 hi = __builtin_elementwise_max(__builtin_convertvector((f2){a1[i][2], a1[i][3]}, h2), z);
    const u32x2 v = {__builtin_bit_cast(unsigned, lo) & c_mask, __builtin_bit_cast(unsigned, hi) & c_mask};
    *reinterpret_cast<u32x2 *>(smem + c_dst + (i >> 1) * 64 + (i & 1) * 32) = v;
  };
  auto c11_slice = [&](int pass, int ks) {
    const int j = pass * 2 + (ks >= 9 ? 1 : 0), t = ks % 9;  // (slot 7 is idle: its groups do not exist and land in the dump)
    if (t == 0) c11_load(j);
    if (t >= 1 && t <= 4) {  // two tiles per slice, the dependent second MFMA of a tile two slices behind its first
      const int m = (t - 1) >> 1, i0 = 2 * ((t - 1) & 1);
      c11_mma(i0, m);
      c11_mma(i0 + 1, m);
    }
    if (t >= 5) {
      c11_store(t - 5);
    }
  };

  // ---- per-lane state of the fused epilogue (nqa_conv_pool.hip) ----
  f32x4 accs[2][2][GPP];  // [pass parity][tile][image]: a pass accumulates into one set while the other set's epilogue runs
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
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < GPP; ++g) accs[1][i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};  // (the phantom epilogue of the first pass reads it)
  float c_vmask = 1.f;
  unsigned c_po_lane = kOOB, c_so_lane = kOOB;
  const bool is15 = l15 == 15, is0 = l15 == 0, even_in = (l15 & 1) == 0 && l15 != 0;
  const size_t nimg = 2 * (size_t)B;
  const unsigned pooled_bytes = (unsigned)(nimg * Ho * Wo * COUT * 2);            // (host-checked < 2^31)
  const unsigned seam_plane = (unsigned)(nimg * strips * Ho * COUT);              // elements; two planes: own | left
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(pooled, 0, pooled_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(seam, 0, 2u * seam_plane * 4u, 0x00020000);
  const unsigned y_pool_off = (unsigned)((size_t)B * Ho * Wo * COUT * 2);
  const unsigned y_seam_off = (unsigned)((size_t)B * strips * Ho * COUT);
  const int ch_lane0 = cg * 32 + 4 * c4;

  float rx = 0.f, ry = 0.f;
  h4 outh[GPP];
  f32x4 seamv[GPP];
  // (the sums and the pool take relu1_2 as the accumulator holds it, NOT rounded to the f16 the unfused path stores the tap
  // in: two conversions per value less in a kernel bound by its vector instructions, and closer to the reference)
  auto pair_values = [&](int p, int par) {  // par: parity of the pass whose accumulators these are
    const int i = p >> 2, e = p & 3;
    rx = fmaxf(accs[par][i][0][e], 0.f);
    ry = fmaxf(accs[par][i][1][e], 0.f);
    if constexpr (ROUND) {
      rx = (float)(_Float16)rx;
      ry = (float)(_Float16)ry;
    }
  };
  auto pair_stats = [&](int p) {
    float dx = rx - piv[p], dy = ry - piv[p];
    if constexpr (RAGGED) {
      dx *= c_vmask;
      dy *= c_vmask;
    }
    s1x[p] += dx;
    s1y[p] += dy;
    s2x[p] = fmaf(dx, dx, s2x[p]);
    s2y[p] = fmaf(dy, dy, s2y[p]);
    sxy[p] = fmaf(dx, dy, sxy[p]);
  };
  auto complete = [&](int i, int g, int e, float s) {
    const float u = U[i][g][e] + s;
    const float a = u + dpp1_row_shr1(u);
    const float pv = a + dpp1_row_shl1(a);
    seamv[g][e] = is15 ? u : pv;
    float val = __builtin_amdgcn_sqrtf(fmaf(pv, 0.0625f, 1e-12f));
    asm volatile("" : "+v"(val));
    outh[g][e] = (_Float16)val;
    U[i][g][e] = s;  // the next window's row -1 (zeroed again between the k loops where a new strip starts)
  };
  auto emit_tile = [&](int i) {
#pragma unroll
    for (int g = 0; g < GPP; ++g) {
      const unsigned po = c_po_lane + (g ? y_pool_off : 0u) + (unsigned)(i * 16) * 2u;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, outh[g]), prsrc, po, 0, 0);
      const unsigned so = c_so_lane + (g ? y_seam_off * 4u : 0u) + (unsigned)(i * 16) * 4u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, seamv[g]), srsrc, so, 0, 0);
    }
  };
  // 18 k-steps carry 32 epilogue slices: two per k-step for the first 16
  auto epi_even_slice = [&](int sl) {
    const int p = sl >> 2, sub = sl & 3, i = p >> 2, e = p & 3;
    if (sub == 0) {
      pair_values(p, 0);
      if (p == 0) n_lane += RAGGED ? c_vmask : 1.f;
      asm volatile("" : "+v"(rx), "+v"(ry));
    }
    if (sub == 1) {
      pair_stats(p);
      asm volatile("" : "+v"(s1x[p]), "+v"(s1y[p]), "+v"(s2x[p]), "+v"(s2y[p]), "+v"(sxy[p]));
    }
    if (sub == 2) {
      const float mx = RAGGED ? rx * c_vmask : rx, my = RAGGED ? ry * c_vmask : ry;
      U[i][0][e] = fmaf(2.f * mx, mx, U[i][0][e]);
      U[i][1][e] = fmaf(2.f * my, my, U[i][1][e]);
      asm volatile("" : "+v"(U[i][0][e]), "+v"(U[i][1][e]));
    }
  };
  auto epi_odd_slice = [&](int sl) {
    const int p = sl >> 2, sub = sl & 3, i = p >> 2, e = p & 3;
    if (sub == 0) {
      pair_values(p, 1);
      if (p == 0) n_lane += RAGGED ? c_vmask : 1.f;
      asm volatile("" : "+v"(rx), "+v"(ry));
    }
    if (sub == 1) {
      pair_stats(p);
      asm volatile("" : "+v"(s1x[p]), "+v"(s1y[p]), "+v"(s2x[p]), "+v"(s2y[p]), "+v"(sxy[p]));
    }
    if (sub == 2) {
      const float mx = RAGGED ? rx * c_vmask : rx;
      complete(i, 0, e, mx * mx);
      asm volatile("" : "+v"(U[i][0][e]), "+v"(seamv[0][e]));
    }
    if (sub == 3) {
      const float my = RAGGED ? ry * c_vmask : ry;
      complete(i, 1, e, my * my);
      asm volatile("" : "+v"(U[i][1][e]), "+v"(seamv[1][e]));
      if (e == 3) emit_tile(i);
    }
  };
  auto epi_even = [&](int ks) {
    if (ks < 16) {
      epi_even_slice(2 * ks);
      epi_even_slice(2 * ks + 1);
    }
  };
  auto epi_odd = [&](int ks) {
    if (ks < 16) {
      epi_odd_slice(2 * ks);
      epi_odd_slice(2 * ks + 1);
    }
  };

  auto reset_stats = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = p >> 2, e = p & 3;
      piv[p] = fmaxf(accs[0][i][0][e], 0.f);  // (pass 0 of the step has just been computed)
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
      for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) r[s] += __shfl_xor(r[s], m, 16);
      }
      // the two half-strip waves of a channel group hold sums of the same channels: rows 2 * block + hs of the partials
      if (l15 == 0) {
        const int c = ch_lane0 + i * 16 + e;
        double *dst = part + (((size_t)n * part_nblk + 2 * blockIdx.x + hs) * COUT + c) * 5;
#pragma unroll
        for (int s = 0; s < 5; ++s) dst[s] = r[s];
      }
    }
  };

  // ---- the k loop of one pass: conv1_2 of one tile row (x | y), the previous pass's epilogue and a piece of the next
  // unit's conv1_1 riding along ----
  int slot_off = 0;
  auto kloop = [&](int pass, auto epi) {
    int rb = slot_off + (pass * G::HWD + hs * 16 + l15) * G::PITCH + (c4 << 4);
    asm volatile("" : "+v"(rb));
    f32x4(&acc)[2][GPP] = accs[pass & 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g) acc[i][g] = bia[i];
    u32x4 bf[PF + 1][GPP];
    auto load_b = [&](int ks, u32x4(&b)[GPP]) {
      const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
      for (int g = 0; g < GPP; ++g)
        b[g] = *reinterpret_cast<const u32x4 *>(smem + rb + (g * G::IMG_BYTES + (ky * G::HWD + kx) * G::PITCH + cc * 64));
    };
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) load_b(ks, bf[ks]);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks + PF < NKS) load_b(ks + PF, bf[(ks + PF) % (PF + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                             __builtin_bit_cast(f16x8, bf[ks % (PF + 1)][g]), acc[i][g], 0, 0, 0);
#ifndef NQA_S1_NO_EPI  // (timing-only ablations, tools/gpu_s1_ablate.sh: results are wrong on purpose)
      epi(ks);
#endif
#ifndef NQA_S1_NO_C11
      c11_slice(pass, ks);
#endif
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto set_ctx = [&](const UC &c, const UC &cn, bool has_next, int pass, bool emit) {
    const int n = c.n, sp = c.sp, ty = c.ty;
    const int s16 = sp * 2 + hs, x0 = s16 * 16, y0 = ty * G::TH;
    if constexpr (RAGGED) c_vmask = (y0 + pass < H && x0 + l15 < W) ? 1.f : 0.f;
    if (pass & 1) {
      const int oy = (y0 >> 1) + (pass >> 1), ox0 = x0 >> 1;
      const bool live = emit && oy < Ho && x0 < W;
      const unsigned pool_row = (((unsigned)n * (unsigned)Ho + (unsigned)oy) * (unsigned)Wo + (unsigned)ox0) * (COUT * 2u);
      const unsigned seam_row = (((unsigned)n * (unsigned)strips + (unsigned)s16) * (unsigned)Ho + (unsigned)oy) * COUT;
      const bool right = s16 + 1 < strips;
      const bool mine = live & even_in & (!RAGGED | (ox0 + (l15 >> 1) < Wo));
      const bool seam_l = live & (is0 | (is15 & right));
      c_po_lane = mine ? pool_row + (unsigned)(((l15 >> 1) * COUT + ch_lane0) * 2) : kOOB;
      c_so_lane = seam_l ? (seam_row + (is15 ? seam_plane + (unsigned)(Ho * COUT) : 0u) + (unsigned)ch_lane0) * 4u : kOOB;
      (void)cn;
      (void)has_next;
    }
  };

  // ---- prologue: raw patches of steps 0 and 1, the halo image of step 0 ----
  __syncthreads();  // the zeroed raw patches
  raw_fetch(uc[0], true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_commit(0);
  raw_fetch(uc[1], 1 < nsteps);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_commit(1);
  __syncthreads();
  c11_set(uc[0], true, 0);
#pragma unroll 1
  for (int j = 0; j < 7; ++j) {
    c11_load(j);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) c11_mma(i, m);
#pragma unroll
    for (int i = 0; i < 4; ++i) c11_store(i);
  }

  int flush_n = -1;
  for (int step = 0; step < nsteps; ++step) {
    const int n = uc[0].n;
    const bool warm = warm0 && step == 0;
    // halo image `step` (written during the previous unit) and raw patch step + 1 (committed at its end) become visible;
    // halo image / raw patch of the other parity are free
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    raw_fetch(uc[2], step + 2 < nsteps);  // lands under this unit's MFMAs
    slot_off = (step & 1) * G::SLOT;
    c11_set(uc[1], step + 1 < nsteps, step + 1);
    const bool fresh = step == (warm0 ? 1 : 0);
    const bool change = !fresh && !warm && n != flush_n;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {  // (unrolled: the pass number, and with it every conv1_1 group index, is static)
      kloop(2 * pp, epi_odd);
      set_ctx(uc[0], uc[1], step + 1 < nsteps, 2 * pp, !warm);
      if (pp == 0) {
        // the previous step's last row is now in U as this tile's row -1: wrong where this tile starts a strip (image top)
        if (uc[0].ty == 0) {
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int g = 0; g < GPP; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) U[i][g][e] = 0.f;
        }
        if (fresh || change) {
          if (change) flush_stats(flush_n);
          if (!warm) {
            reset_stats();
            flush_n = n;
          }
        }
      }
      kloop(2 * pp + 1, epi_even);
      set_ctx(uc[0], uc[1], step + 1 < nsteps, 2 * pp + 1, !warm);
    }
    // the raw pixels of step + 2 were requested a whole unit ago; the only younger operations are this unit's 2 * NST
    // stores (and, rarely, a flush's): retire the pixels without waiting for the stores
    if (change)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NST) : "memory");
    raw_commit(step & 1);
    uc[0] = uc[1];
    uc[1] = uc[2];
    uc[2] = next_uc(uc[2]);
  }
#pragma unroll
  for (int sl = 0; sl < 32; ++sl) epi_odd_slice(sl);
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
static thread_local int g_round_tap1 = 0;   // bit 8 (256): the fused stage 1 rounds relu1_2 to f16 before its sums and pool (tests)
void set_fuse_stage1(int on) { g_fuse_stage1 = on & 1; g_round_tap1 = (on >> 1) & 1; }

bool conv1_pool_fusable(int B, int H, int W, int blob_prec) {
  if (!g_fuse_stage1 || blob_prec != NQA_PREC_F16 || W < 16 || H < 4) return false;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, strips = cdiv(W, 16);
  if ((size_t)H * W * 3 * 4 >= (1ull << 31)) return false;
  if (2ull * B * Ho * Wo * 64 * 2 >= (1ull << 31)) return false;
  if (2ull * 2 * B * strips * Ho * 64 * 4 >= (1ull << 31)) return false;
  return true;
}

template <bool RAGGED, bool ROUND>
static int launch_conv1_pool(const float *x, const float *y, int B, int H, int W, const char *packed, void *pooled, float *seam,
                             double *part, hipStream_t st) {
  typedef S1Geom G;
  static std::atomic<bool> attr_done_dev[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<bool> &attr_done = attr_done_dev[dev & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_pool_kernel<RAGGED, ROUND>), hipFuncAttributeMaxDynamicSharedMemorySize,
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
    conv1_pool_kernel<RAGGED, ROUND><<<grid, 256, G::LDS, st>>>(x, y, B, packed + layer0_m16_offset(NQA_PREC_F16), b1,
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
  if (g_round_tap1)
    return ragged ? launch_conv1_pool<true, true>(x, y, B, H, W, p, pooled, seam, part, st)
                  : launch_conv1_pool<false, true>(x, y, B, H, W, p, pooled, seam, part, st);
  return ragged ? launch_conv1_pool<true, false>(x, y, B, H, W, p, pooled, seam, part, st)
                : launch_conv1_pool<false, false>(x, y, B, H, W, p, pooled, seam, part, st);
}

}  // namespace nqa
