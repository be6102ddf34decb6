// conv2_2 + bias + ReLU + L2-pool + the five DISTS statistics sums in ONE kernel (gfx950, 16-bit modes).
//
// The DISTS path never needs the full-resolution tap relu2_2 again once it has been pooled (the next stage's input,
// DISTS_pt.py:22-25,94-95) and summed (the per-channel means / variances / covariance behind S1, S2,
// DISTS_pt.py:130-142): until round 3 the conv wrote it (2.1 GB per B=8 1080p step), and pool_stats_kernel read it back.
// Here the tap only ever exists in registers:
//   * conv3x3_regw128_kernel's loop (weights in registers, ONE wave per SIMD, halo patch by LDS-DMA one tile ahead), on
//     a tile of 4 rows x 16 columns of an image PAIR: the two 16-pixel MFMA column groups of a pass are the x image and
//     the y image at the same position (x at patch columns [0,18), y at [24,42): the same chunk-swizzle phase, so the
//     second group is an immediate offset), hence a lane holds x and y of one (pixel, 4 channels) side by side and
//     sum x*y is a lane-local product -- nothing is kept from one image for the other;
//   * statistics: per lane SHIFTED moments (pivot = the channel's first sample after a flush; see nqa_pool_stats.hip's
//     ShiftedMoments for why) of the f16-ROUNDED activations, i.e. of exactly the values the unfused path would have
//     stored and read back; converted to fp64 raw sums and folded over the 16 pixel lanes when the block leaves an
//     image pair (and at its end), one partial row per (pair, block) for finalize_kernel;
//   * L2-pool = separable (1,2,1)/4 x (1,2,1)/4 of the squares, stride 2, zero pad 1: the VERTICAL part is lane-local
//     (a wave owns all four rows of its channels: U = s[2j-1] + 2 s[2j] + s[2j+1], the row above the tile carried in
//     registers from the previous tile of the strip -- blocks walk strips of tiles top to bottom), the HORIZONTAL part
//     is two DPP adds inside a 16-lane row (a = U + U[lane-1]; P = a + a[lane+1], taken at the even lanes);
//   * the one pooled column per strip whose window reaches into the strip to the left (lane 0) leaves as an fp32
//     partial sum, the left strip's contribution to it (its lane 15's U) as another; pool_seam_kernel adds the two,
//     takes the square root and writes that column (1/8 of the pooled map).  Everything else is written once, final.
//   * the epilogue of pass p (bias, ReLU, rounding, sums, pool) is issued INSIDE the k loop of pass p + 1, a few
//     vector instructions behind each MFMA (one wave per SIMD: there is no partner wave to run it under), so the matrix
//     pipe never waits for it.
// A-DISTS needs the taps themselves and keeps the unfused kernels.
#include <atomic>

#include "nqa_common.h"

namespace nqa {

typedef __attribute__((address_space(3))) void lds_void_p;

__device__ __host__ static inline int swz16(int r) { return ((r >> 2) & 1) * 2; }  // (lds_swz<true> of nqa_conv.hip)

__device__ static inline float dpp_row_shr1(float v) {  // lane l of each 16-lane row reads lane l-1; lane 0 reads 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ static inline float dpp_row_shl1(float v) {  // lane l reads lane l+1; lane 15 reads 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}

struct PoolGeom {
  static constexpr int CIN = 128, NCC = 4, NKS = NCC * 9, TH = 4, TWI = 16;
  static constexpr int YCOL = 24, HWD = YCOL + TWI + 2, NQ = (TH + 2) * HWD;  // 42 patch columns, 252 halo pixels
  static constexpr int CH_ITEMS = 1024, CH_BYTES = CH_ITEMS * 16, SLOT = NCC * CH_BYTES, LDS = 2 * SLOT;
  static_assert(NQ * 4 <= CH_ITEMS, "patch does not fit the DMA rounds");
};

// NTERM = 2 (the mixed modes): a wave's two fragment sets are the (hi, lo) parts of 16 channels' weights, a block covers
// 64 output channels (Cout / 64 channel tiles per unit position).  RAGGED: H % 4 or W % 16 (out-of-image pixels of a
// tile are masked out of the pool and the sums; the instances for full tiles carry no masking at all).
// OUT_SPLIT: the pooled map leaves as split16 records (it feeds an f32s stage: NQA_PREC_F32M2's boundary).
template <int NTERM, bool RAGGED, bool OUT_SPLIT>
__global__ __launch_bounds__(256, 1) void conv3x3_regw128_pool_kernel(
    const _Float16 *__restrict__ in, const char *__restrict__ wreg, const float *__restrict__ bias,
    char *__restrict__ pooled, float *__restrict__ seam, double *__restrict__ part, int B, int H, int W, int Cout,
    int strips, int rows, int total_units, int part_nblk) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 T;
  typedef PoolGeom G;
  constexpr int PF = RAGGED ? 1 : 3, GPP = 2;  // (the masked instances trade a fragment set for the mask's registers: no spill)
  constexpr int NI = 2 / NTERM;        // 16-channel tiles a wave OWNS (NTERM = 2: its two fragment sets are one tile's hi / lo)
  constexpr int NP = NI * 4;           // (x, y) value pairs per lane and pass: (tile i, channel e)
  constexpr int KPP = 32 / NP;         // k-steps of the next pass's loop that carry one pair's epilogue
  constexpr int BC = 128 / NTERM;      // output channels per block
  constexpr int NST = NI * 2 * (OUT_SPLIT ? 3 : 2);  // stores per wave and pooled row (per even-pass k loop)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int nct = Cout / BC;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;

  // ---- this block's run of units: contiguous in [pair][channel tile][strip][row] order; the blocks that share an XCD
  // (ids equal mod 8) own adjacent runs, i.e. neighbouring strips, whose two shared halo columns then hit one L2 ----
  const int nblk = gridDim.x;
  int run;
  {
    const int qq = nblk >> 3, rr = nblk & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    run = nblk < 8 ? (int)blockIdx.x : (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
  }
  const int u_lo = (int)((long)total_units * run / nblk), u_hi = (int)((long)total_units * (run + 1) / nblk);
  if (u_lo >= u_hi) return;  // (block-uniform)
  // a run that starts inside a strip first recomputes the LAST TWO rows of the tile above (its row 3 is the pool's row
  // -1 of the first real tile): a warm-up step whose outputs and sums are dropped
  const bool warm0 = (u_lo % rows) != 0;
  const int nsteps = (u_hi - u_lo) + (warm0 ? 1 : 0);
  auto coords = [&](int step, int &n, int &ct, int &sx, int &ty) {
    int u = u_lo + step - (warm0 ? 1 : 0);
    ty = u % rows;
    u /= rows;
    sx = u % strips;
    u /= strips;
    ct = u % nct;
    n = u / nct;
  };

  // ---- weights ----
  u32x4 wf[2][G::NKS];
  float bia[NI][4];
  int cur_ct = -1;
  auto load_weights = [&](int ct) {
    const int g = ct * 4 + wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < G::NKS; ++ks)
        wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)g * 2 + i) * G::NKS + ks) * 64 + lane) * 16);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) bia[i][e] = bias[NTERM == 2 ? g * 16 + 4 * c4 + e : g * 32 + i * 16 + 4 * c4 + e];
    cur_ct = ct;
  };
  const float winv = NTERM == 2 ? bias[Cout] : 1.f;

  // ---- halo DMA plan: 4 rounds of 256 items (16 B) per 32-channel chunk; item j = quarter (j&3)^swz(col) of patch
  // pixel q = j>>2 = (row q/42, col q%42); cols [0,18) are the x image, [24,42) the y image (B images further on) ----
  int p_hy[4], p_hx[4], p_c[4];
  unsigned p_img[4];
  const unsigned img_bytes = (unsigned)H * (unsigned)W * (unsigned)G::CIN * 2u;
  const unsigned pair_off = (unsigned)B * img_bytes;  // (host-checked: (B + 1) * img_bytes < 2^31)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = r * 256 + tid, q = j >> 2;
    const int hy = q / G::HWD, col = q - hy * G::HWD;
    const bool live = q < G::NQ && (col < G::TWI + 2 || col >= G::YCOL);
    p_hy[r] = live ? hy : -100000;
    p_hx[r] = col >= G::YCOL ? col - G::YCOL : col;
    p_img[r] = col >= G::YCOL ? pair_off : 0u;
    p_c[r] = (j & 3) ^ swz16(col);
  }
  int tap_base[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) tap_base[kx] = (l15 + kx) * 64 + ((c4 ^ swz16(l15 + kx)) << 4);
  const unsigned kOOB = 0x80000000u;
  auto issue_halo = [&](int step, int slot_idx) {
    char *slot = smem + slot_idx * G::SLOT;
    const bool real = step < nsteps;
    int n = 0, ct = 0, sx = 0, ty = 0;
    if (real) coords(step, n, ct, sx, ty);
    const int x0 = sx * G::TWI, y0 = ty * G::TH;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(in + (size_t)n * H * W * G::CIN), 0, pair_off + img_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gy = y0 - 1 + p_hy[r], gx = x0 - 1 + p_hx[r];
      const bool ok = real && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? p_img[r] + (unsigned)(((gy * W + gx) * G::CIN + p_c[r] * 8) * 2) : kOOB;
#pragma unroll
      for (int cc = 0; cc < G::NCC; ++cc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_p *)(slot + cc * G::CH_BYTES + r * 4096 + wave * 1024), 16,
                                                 off, cc * 64, 0, 0);
      if (real && !ok && p_hy[r] >= 0) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int cc = 0; cc < G::NCC; ++cc) *reinterpret_cast<u32x4 *>(slot + cc * G::CH_BYTES + (r * 256 + tid) * 16) = z;
      }
    }
  };

  // ---- per-lane state of the fused epilogue ----
  f32x4 acc[2][GPP], pacc[2][GPP];      // this pass's accumulators; the previous pass's, whose epilogue is under way
  float U[NI][GPP][4];                  // vertical pool sums in the making (value index = (tile i, image g, channel e))
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
    for (int g = 0; g < GPP; ++g) pacc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // context of the pass whose epilogue is under way (block-uniform scalars but for vmask)
  float c_keep = 0.f;        // odd passes: what the completed row leaves as the next window's row -1 (1: the tile row below
                             // follows in this strip; 0: image top / another strip next)
  float c_vmask = 1.f;       // RAGGED: 1 where the pass's pixel of this lane lies inside the image
  unsigned c_po_lane = kOOB;  // this lane's byte offset in `pooled` (x image, channel tile 0 of the wave), kOOB: nothing to store
  unsigned c_so_lane = kOOB;  // ... in the seam planes (lanes 0 and 15 only)
  const bool is15 = l15 == 15, is0 = l15 == 0, even_in = (l15 & 1) == 0 && l15 != 0;
  const unsigned pool_esz = OUT_SPLIT ? 4u : 2u;
  const size_t nimg = 2 * (size_t)B;
  const unsigned pooled_bytes = (unsigned)(nimg * Ho * Wo * Cout * pool_esz);      // (host-checked < 2^31)
  const unsigned seam_plane = (unsigned)(nimg * strips * Ho * Cout);               // elements; two planes: own | left
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(pooled, 0, pooled_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(seam, 0, 2u * seam_plane * 4u, 0x00020000);
  const unsigned y_pool_off = (unsigned)((size_t)B * Ho * Wo * Cout * pool_esz);
  const unsigned y_seam_off = (unsigned)((size_t)B * strips * Ho * Cout);
  int ch_lane[NI];  // first channel of this lane's 4 in tile i, relative to the block's channel tile
#pragma unroll
  for (int i = 0; i < NI; ++i) ch_lane[i] = (NTERM == 2 ? wave * 16 : wave * 32 + i * 16) + 4 * c4;

  // ---- the epilogue in slices.  `ks` is a compile-time constant after unrolling; pair p = (tile i, channel e) ----
  float rx = 0.f, ry = 0.f;                     // the pair's rounded activations, between its slices
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  h4 outh[GPP];                                 // a tile's four pooled channels per image, until its store
  f32x4 seamv[GPP];
  auto pair_values = [&](int p) {               // slice 0 of a pair: bias, ReLU, rounding to the tap's f16
    const int i = p >> 2, e = p & 3;
    float ax, ay;
    if constexpr (NTERM == 2) {
      ax = (pacc[0][0][e] + pacc[1][0][e]) * winv;
      ay = (pacc[0][1][e] + pacc[1][1][e]) * winv;
    } else {
      ax = pacc[i][0][e];
      ay = pacc[i][1][e];
    }
    rx = (float)(_Float16)fmaxf(ax + bia[i][e], 0.f);
    ry = (float)(_Float16)fmaxf(ay + bia[i][e], 0.f);
  };
  auto pair_stats = [&](int p) {                // slice 1: the five shifted sums
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
  auto complete = [&](int i, int g, int e, float s) {  // odd passes: the value's pooled row is complete
    const float u = U[i][g][e] + s;
    const float a = u + dpp_row_shr1(u);
    const float pv = a + dpp_row_shl1(a);
    seamv[g][e] = is15 ? u : pv;
    float val = __builtin_amdgcn_sqrtf(fmaf(pv, 0.0625f, 1e-12f));  // (v_sqrt_f32: 1 ulp, then rounded to half)
    asm volatile("" : "+v"(val));                                   // (keeps the root in this slice, not bunched at the store)
    outh[g][e] = (_Float16)val;
    U[i][g][e] = s * c_keep;
  };
  auto emit_tile = [&](int i) {                 // after a tile's four channels: pooled pixels and seam partials leave
    // (c_po_lane / c_so_lane: this lane's byte offsets for tile 0 of the x image, or kOOB -- set_ctx; an out-of-range
    // offset stays out of range under the small additions below, so every wave issues the same stores whatever it emits)
#pragma unroll
    for (int g = 0; g < GPP; ++g) {
      const unsigned po = c_po_lane + (g ? y_pool_off : 0u) + (unsigned)(i * 16) * pool_esz;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, outh[g]), prsrc, po, 0, 0);
      // lane 0: this strip's own partial of pooled column x0/2 -> plane 0; lane 15: its column for the strip to the right -> plane 1
      const unsigned so = c_so_lane + (g ? y_seam_off * 4u : 0u) + (unsigned)(i * 16) * 4u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, seamv[g]), srsrc, so, 0, 0);
    }
  };
  // Each slice ends in an empty `asm volatile` on what it produced: instruction selection orders side-effecting nodes
  // only, so without these anchors a slice of pure arithmetic floats to the END of the (fully unrolled) loop body, out of
  // the MFMAs' shadow -- the first build ran the whole even-pass epilogue behind its k loop (found in the ISA).
  auto epi_even = [&](int ks) {
    const int p = ks / KPP, sub = ks % KPP;
    if (p >= NP) return;
    const int i = p >> 2, e = p & 3;
    if (sub == 0) {
      pair_values(p);
      if (p == 0) n_lane += RAGGED ? c_vmask : 1.f;  // samples behind this lane's sums (one per pass)
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
  // odd passes (rows 1, 3): U += s completes pooled row j = (row - 1) / 2; runs inside the following even pass's k loop
  auto epi_odd = [&](int ks) {
    const int p = ks / KPP, sub = ks % KPP;
    if (p >= NP) return;
    const int i = p >> 2, e = p & 3;
    if (sub == 0) {
      pair_values(p);
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

  // ---- statistics: flush the lane sums of pair `n`, channel tile `ct` (fp64 raw sums over the 16 pixel lanes) ----
  auto reset_stats = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = p >> 2, e = p & 3;
      // the pivot: this lane's activation of the pass whose accumulators have just been set aside
      float a0;
      if constexpr (NTERM == 2) a0 = (pacc[0][0][e] + pacc[1][0][e]) * winv; else a0 = pacc[i][0][e];
      piv[p] = fmaxf(a0 + bia[i][e], 0.f);
      s1x[p] = s1y[p] = s2x[p] = s2y[p] = sxy[p] = 0.f;
    }
    n_lane = 0.f;
  };
  auto flush_stats = [&](int n, int ct) {
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
      if (l15 == 0) {
        const int c = ct * BC + ch_lane[i] + e;
        double *dst = part + (((size_t)n * part_nblk + blockIdx.x) * Cout + c) * 5;
#pragma unroll
        for (int s = 0; s < 5; ++s) dst[s] = r[s];
      }
    }
  };

  // ---- the k loop of one pass (one tile row, x | y) with the previous pass's epilogue riding along ----
  int slot_off = 0;
  auto kloop = [&](int pass, auto epi) {
    int rowb[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) rowb[kx] = tap_base[kx] + slot_off + pass * (G::HWD * 64);
    asm volatile("" : "+v"(rowb[0]), "+v"(rowb[1]), "+v"(rowb[2]));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 bf[PF + 1][GPP];
    auto load_b = [&](int ks, u32x4(&b)[GPP]) {
      const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
      for (int g = 0; g < GPP; ++g)
        b[g] = *reinterpret_cast<const u32x4 *>(smem + rowb[kx] + (cc * G::CH_BYTES + ky * (G::HWD * 64) + g * (G::YCOL * 64)));
    };
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) load_b(ks, bf[ks]);
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
      if (ks + PF < G::NKS) load_b(ks + PF, bf[(ks + PF) % (PF + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                             __builtin_bit_cast(f16x8, bf[ks % (PF + 1)][g]), acc[i][g], 0, 0, 0);
      epi(ks);
      // one MFMA, then two of the epilogue's vector instructions, four times: the epilogue hides in the MFMAs' shadow
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g) pacc[i][g] = acc[i][g];
  };
  // the context of the pass just computed (its epilogue runs next)
  auto set_ctx = [&](int step, int pass, bool emit) {
    int n, ct, sx, ty;
    coords(step, n, ct, sx, ty);
    const int x0 = sx * G::TWI, y0 = ty * G::TH;
    if constexpr (RAGGED) c_vmask = (y0 + pass < H && x0 + l15 < W) ? 1.f : 0.f;
    if (pass & 1) {
      const int oy = (y0 >> 1) + (pass >> 1), ox0 = x0 >> 1;
      const bool live = emit && oy < Ho;
      const unsigned pool_row = (unsigned)((((size_t)n * Ho + oy) * Wo + ox0) * Cout + ct * BC) * pool_esz;
      const unsigned seam_row = (unsigned)((((size_t)n * strips + sx) * Ho + oy) * Cout + ct * BC);
      const bool right = sx + 1 < strips;
      // (bitwise, not short-circuit: no branches)
      const bool mine = live & even_in & (!RAGGED | (ox0 + (l15 >> 1) < Wo));
      const bool seam_l = live & (is0 | (is15 & right));
      c_po_lane = mine ? pool_row + (unsigned)(((l15 >> 1) * Cout + ch_lane[0]) * (int)pool_esz) : kOOB;
      c_so_lane = seam_l ? (seam_row + (is15 ? seam_plane + (unsigned)(Ho * Cout) : 0u) + (unsigned)ch_lane[0]) * 4u : kOOB;
      if (pass == 1) {
        c_keep = 1.f;
      } else {  // row 3: is the next step the tile below, in this strip?
        int n2, ct2, sx2, ty2;
        c_keep = 0.f;
        if (step + 1 < nsteps) {
          coords(step + 1, n2, ct2, sx2, ty2);
          c_keep = (n2 == n && ct2 == ct && sx2 == sx && ty2 == ty + 1) ? 1.f : 0.f;
        }
      }
    }
  };

  issue_halo(0, 0);
  int flush_n = -1, flush_ct = 0;   // the (pair, channel tile) whose sums the lanes currently hold (-1: none worth keeping)
  bool prev_warm = false;
  for (int step = 0; step < nsteps; ++step) {
    int n, ct, sx, ty;
    coords(step, n, ct, sx, ty);
    const bool warm = warm0 && step == 0;
    if (ct != cur_ct) {  // (block-uniform; compiler-tracked loads, retired with a wait the compiler can see: nqa_conv.hip)
      load_weights(ct);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    // halo `step` has landed: the only younger operations are the previous step's stores
    if (step == 0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (prev_warm)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NST) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NST) : "memory");
    issue_halo(step + 1, (step + 1) & 1);
    slot_off = (step & 1) * G::SLOT;
    // what to do with the lane sums once the epilogue of the PREVIOUS step's last pass has run (inside this step's
    // first even pass): nothing, start afresh (first real step), or hand the finished pair over and start afresh
    const bool fresh = step == (warm0 ? 1 : 0);
    const bool change = !fresh && !warm && (n != flush_n || ct != flush_ct);
#pragma unroll 1
    for (int pp = warm ? 1 : 0; pp < 2; ++pp) {
      kloop(2 * pp, epi_odd);
      set_ctx(step, 2 * pp, !warm);
      if (pp == (warm ? 1 : 0) && (fresh || change)) {
        if (change) {
          flush_stats(flush_n, flush_ct);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (rare; keeps the counted waits of the ring exact)
        }
        if (!warm) {
          reset_stats();
          flush_n = n;
          flush_ct = ct;
        }
      }
      kloop(2 * pp + 1, epi_even);
      set_ctx(step, 2 * pp + 1, !warm);
    }
    prev_warm = warm;
  }
  // ---- drain: the last pass's epilogue, then this block's last sums ----
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) epi_odd(ks);
  if (flush_n >= 0) flush_stats(flush_n, flush_ct);
#endif
}

// The pooled column of every strip that neighbours another strip: own partial + the left strip's contribution.
template <bool OUT_SPLIT>
__global__ __launch_bounds__(256) void pool_seam_kernel(const float *__restrict__ seam, char *__restrict__ pooled, int nimg,
                                                        int strips, int Ho, int Wo, int C, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // one thread = 4 channels of one seam pixel
  if (idx >= total) return;
  const int c4n = C >> 2;
  const int cq = (int)(idx % c4n);
  long t = idx / c4n;
  const int oy = (int)(t % Ho);
  t /= Ho;
  const int sx = (int)(t % strips);
  const int img = (int)(t / strips);
  const size_t so = ((((size_t)img * strips + sx) * Ho + oy) * C + cq * 4);
  const size_t plane = (size_t)nimg * strips * Ho * C;
  f32x4 own = *reinterpret_cast<const f32x4 *>(seam + so);
  if (sx > 0) {
    const f32x4 left = *reinterpret_cast<const f32x4 *>(seam + plane + so);
    own += left;
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_sqrtf(fmaf(own[e], 0.0625f, 1e-12f));
  char *pix = pooled + (((size_t)img * Ho + oy) * Wo + sx * 8) * C * (OUT_SPLIT ? 4 : 2);
  if constexpr (OUT_SPLIT) {
    store_split4(pix, cq * 4, v[0], v[1], v[2], v[3]);
  } else {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<h4 *>(pix + cq * 8) = o;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------
// the seam pass behind a fused conv + pool kernel (f16 pooled map; also nqa_conv1_pool.hip's)
int pool_seam_finish(const float *seam, void *pooled, int nimg, int strips, int Ho, int Wo, int C, hipStream_t st) {
  const long total = (long)nimg * strips * Ho * (C / 4);
  TimedLaunch t(NQA_K_SEAM, st);
  pool_seam_kernel<false><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(seam, static_cast<char *>(pooled), nimg, strips, Ho, Wo, C,
                                                                             total);
  return check_launch("pool_seam");
}

static int pool_num_cus() {
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

static thread_local int g_fuse_taps = 1;  // nqa_set_conv_variant bit 6 (64): 0 = the unfused path, for A/B runs
void set_fuse_taps(int on) { g_fuse_taps = on; }

// Can conv layer `layer` of a B-pair batch of H x W maps (its input's size) take the fused form?
bool conv_pool_fusable(int layer, int B, int H, int W, int blob_prec, int kprec) {
  if (!g_fuse_taps || layer != 3 || kprec != NQA_PREC_F16 || W < 16 || H < 4) return false;
  if (!(blob_prec == NQA_PREC_F16 || is_mixed(blob_prec))) return false;
  const size_t img_in = (size_t)H * W * 128 * 2;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, strips = cdiv(W, 16);
  const bool split_out = is_mixed(blob_prec) && stage_prec(blob_prec, 2) == NQA_PREC_F32S;
  if (split_out) return false;  // (the split16 boundary keeps the unfused pass for now)
  if ((B + 1) * img_in >= (1ull << 31)) return false;                              // 32-bit offsets of the DMA plan
  if (2ull * B * Ho * Wo * 128 * (split_out ? 4 : 2) >= (1ull << 31)) return false;  // ... of the pooled stores
  if (2ull * 2 * B * strips * Ho * 128 * 4 >= (1ull << 31)) return false;           // ... of the seam planes
  return true;
}
int conv_pool_part_blocks() { return NQA_FUSED_PART_BLOCKS; }
// bytes of the two seam planes of a (B pairs, H x W input map, C channels) fused tap
size_t conv_pool_seam_bytes(int B, int H, int W, int C) {
  return align_up(2ull * 2 * B * cdiv(W, 16) * ((H + 1) / 2) * C * 4, 256);
}

template <int NTERM, bool RAGGED>
static int launch_conv_pool(const void *in, int B, int H, int W, int layer, const char *packed, int blob_prec, void *pooled,
                            float *seam, double *part, hipStream_t st) {
  typedef PoolGeom G;
  static std::atomic<bool> attr_done_dev[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<bool> &attr_done = attr_done_dev[dev & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_regw128_pool_kernel<NTERM, RAGGED, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) {
      set_error("conv_pool: cannot raise the dynamic LDS limit to %d bytes", G::LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = pool_num_cus();
  if (!cus) {
    set_error("conv_pool: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int cout = kConvs[layer].cout, nct = cout / (128 / NTERM);
  const int strips = cdiv(W, G::TWI), rows = cdiv(H, G::TH);
  const long units = (long)B * nct * strips * rows;
  int grid = (int)(units < cus ? units : cus);
  if (grid > NQA_FUSED_PART_BLOCKS) grid = NQA_FUSED_PART_BLOCKS;
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, blob_prec));
  // every (pair, block) partial row this launch does not write must read as zero
  if (hipMemsetAsync(part, 0, (size_t)B * NQA_FUSED_PART_BLOCKS * cout * 5 * sizeof(double), st) != hipSuccess) {
    set_error("conv_pool: memset of the statistics partials failed");
    return NQA_E_LAUNCH;
  }
  {
    TimedLaunch t(NQA_K_CONV, st);
    conv3x3_regw128_pool_kernel<NTERM, RAGGED, false><<<grid, 256, G::LDS, st>>>(
        reinterpret_cast<const _Float16 *>(in), packed + regw_offset(layer, blob_prec), bias, static_cast<char *>(pooled), seam,
        part, B, H, W, cout, strips, rows, (int)units, NQA_FUSED_PART_BLOCKS);
    const int rc = check_launch("conv3x3_regw128_pool");
    if (rc) return rc;
  }
  return pool_seam_finish(seam, pooled, 2 * B, strips, (H + 1) / 2, (W + 1) / 2, cout, st);
}

// conv layer 3 (conv2_2) of the 2B-image batch `in` (x images [0,B), y images [B,2B), NHWC f16, 128 channels) ->
// pooled map (2B images) + statistics partials of the B pairs; the tap is never written.
int conv_pool_stats_fused(const void *in, int B, int H, int W, int layer, const void *packed, int blob_prec, void *pooled,
                          float *seam, double *part, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  const bool ragged = (H % 4) != 0 || (W % 16) != 0;
  if (is_mixed(blob_prec))
    return ragged ? launch_conv_pool<2, true>(in, B, H, W, layer, p, blob_prec, pooled, seam, part, st)
                  : launch_conv_pool<2, false>(in, B, H, W, layer, p, blob_prec, pooled, seam, part, st);
  return ragged ? launch_conv_pool<1, true>(in, B, H, W, layer, p, blob_prec, pooled, seam, part, st)
                : launch_conv_pool<1, false>(in, B, H, W, layer, p, blob_prec, pooled, seam, part, st);
}

}  // namespace nqa
