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
//                       (nqa_pack_vgg_weights) one kernel row (3 taps) at a time.  Staging is
//                       LDS-DMA, double-buffered, one barrier per stage.
//
// LDS images are rows of 64 bytes (one pixel or one output channel, 4 chunks of 16 B).
// Chunk c of row r sits at position c ^ lds_swz(r), chosen per MFMA shape so that every
// ds_read_b128 lane group covers all 16 slots of the 256-byte bank row from any base row:
// (r>>2)&3 for the 32x32x16 reads (32 consecutive rows, one chunk), 2*((r>>2)&1) for the
// 16x16x32 reads (16 consecutive rows, all four chunks) -- cdna guide section 2 / T2.
//
// Three more kernels live here: conv1_fused_kernel and conv1_tile_kernel (stage 1 without its HBM
// intermediate, two forms) and the precision variants of the implicit GEMM (f32: exact-f32 MFMA;
// f32s: split-f16 products on split16 activations; 16-bit: v_mfma_f32_16x16x32).
#include <atomic>

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
  // One thread per output pixel (weights are wave-uniform -> scalar operands of the FMAs).  The
  // 64 channels of 256 consecutive pixels are contiguous in NHWC, so results go through an LDS
  // tile and leave as fully coalesced 16-byte stores, 128 bytes of channels per pass.
  constexpr int HC = 128 / (int)sizeof(T);  // channels per pass: 64 (16-bit) or 32 (f32)
  constexpr int PITCH = 128 + 16;           // row pitch in bytes; +16 spreads ds_write_b128 over the banks
  __shared__ __attribute__((aligned(16))) char tile[256 * PITCH];
  const int HW = H * W;
  const int tid = threadIdx.x;
  const int pix0 = blockIdx.x * 256, pix = pix0 + tid;
  const int n = blockIdx.y;
  const bool live = pix < HW;
  const int py = live ? pix / W : 0, px = live ? pix - py * W : 0;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  float in[27];
  const float *xi = x + (size_t)n * 3 * HW;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int gy = py + ky - 1, gx = px + kx - 1;
      const bool ok = live && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (ok) v = (xi[(size_t)c * HW + gy * W + gx] - mean[c]) / sd[c];
        in[(ky * 3 + kx) * 3 + c] = v;
      }
    }
  }
  char *o = reinterpret_cast<char *>(out + ((size_t)n * HW + pix0) * 64);
  const int npix = min(256, HW - pix0);
#pragma unroll 1
  for (int half = 0; half < 64 / HC; ++half) {
    // groups of 16 output channels keep accumulators + inputs under 64 VGPRs
#pragma unroll 1
    for (int g = 0; g < HC / 16; ++g) {
      const int c0 = half * HC + g * 16;
      // channel pairs as packed floats: v_pk_fma_f32 with the two weights as one scalar pair
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      f32x2 acc2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc2[j] = *reinterpret_cast<const f32x2 *>(bias + c0 + 2 * j);
#pragma unroll
      for (int k = 0; k < 27; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc2[j] = in[k] * *reinterpret_cast<const f32x2 *>(w + k * 64 + c0 + 2 * j) + acc2[j];
      }
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = acc2[j >> 1][j & 1];
      T *row = reinterpret_cast<T *>(tile + tid * PITCH) + g * 16;
      if constexpr (P::SPLIT) {  // one split16 record (64 B) per 16 channels, same bytes per pass as float
#pragma unroll
        for (int j = 0; j < 16; j += 4)
          store_split4(reinterpret_cast<char *>(row), j, fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f),
                       fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f));
      } else if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          f32x4 v = {fmaxf(acc[j], 0.f), fmaxf(acc[j + 1], 0.f), fmaxf(acc[j + 2], 0.f), fmaxf(acc[j + 3], 0.f)};
          *reinterpret_cast<f32x4 *>(row + j) = v;
        }
      } else {
        typedef __attribute__((ext_vector_type(8))) T t8;
#pragma unroll
        for (int j = 0; j < 16; j += 8) {
          t8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = P::from_f(fmaxf(acc[j + e], 0.f));
          *reinterpret_cast<t8 *>(row + j) = v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = r * 256 + tid, p = i >> 3, ch = i & 7;
      if (p < npix)
        *reinterpret_cast<u32x4 *>(o + (size_t)p * 64 * sizeof(T) + half * 128 + ch * 16) =
            *reinterpret_cast<const u32x4 *>(tile + p * PITCH + ch * 16);
    }
    if (half + 1 < 64 / HC) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// implicit-GEMM conv, layers 1..12
// ---------------------------------------------------------------------------------
// Block tile = BN output channels x MP output pixels (a TH x TW patch of one image), split
// over WAVES_N x WAVES_M waves, each owning WN_T x WM_T MFMA tiles of 32x32.
// A "stage" is one kernel row (3 taps) of one 64-byte channel chunk: 3*BN weight rows.
// Both LDS images are filled by LDS-DMA (buffer_load_dwordx4 ... lds: per-lane source
// offset, wave-linear destination), double-buffered, so the loop has ONE barrier per
// stage: vmcnt(0) + barrier retires stage s while stage s+1 is already in flight.
typedef __attribute__((address_space(3))) void lds_void_t;

#ifdef NQA_STAMPS  // diagnostic build: per-segment shader-cycle sums of the stage loop (never shipped)
__device__ unsigned long long g_stamps[8];
__device__ inline unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define NQA_STAMP(var) const unsigned long long var = stamp()
#define NQA_STAMP_ADD(seg, a, b) seg_sum[seg] += (b) - (a)
#else
#define NQA_STAMP(var)
#define NQA_STAMP_ADD(seg, a, b)
#endif

template <int WAVES_N, int WAVES_M, int WN_T, int WM_T, int TW>
struct ConvGeom {
  static constexpr int THREADS = 64 * WAVES_N * WAVES_M;
  static constexpr int BN = WAVES_N * WN_T * 32;      // output channels per block
  static constexpr int NSUB = BN / 64;                // 64-channel weight sub-slabs
  static constexpr int MP = WAVES_M * WM_T * 32;      // output pixels per block
  static constexpr int TH = MP / TW;                  // tile rows
  static constexpr int HW_ = TW + 2, HH_ = TH + 2;    // halo extent
  static constexpr int NQ = HW_ * HH_;                // halo pixels
  static constexpr int A_ITEMS = NQ * 4;              // 16-byte chunks
  static constexpr int A_ITEMS_PAD = (A_ITEMS + 63) / 64 * 64;
  static constexpr int A_BYTES = A_ITEMS_PAD * 16;
  static constexpr int A_ROUNDS = (A_ITEMS_PAD + THREADS - 1) / THREADS;
  static constexpr int SUB_STAGE_ITEMS = 3 * 64 * 4;  // one sub-slab, one kernel row: 12 KB
  static constexpr int W_ITEMS = NSUB * SUB_STAGE_ITEMS;
  static constexpr int W_BYTES = W_ITEMS * 16;
  static constexpr int W_ROUNDS = W_ITEMS / THREADS;
  static constexpr int LDS_BYTES = 2 * (A_BYTES + W_BYTES);
  static_assert(BN % 64 == 0 && W_ITEMS % THREADS == 0 && MP % TW == 0, "bad conv geometry");
};

// LDS row swizzle: chunk c of row r sits at position c ^ lds_swz(r).  The 32x32x16 MFMA reads one
// chunk of 32 consecutive rows per instruction (rows spread by (r>>2)&3); the 16x16x32 MFMA reads
// all four chunks of 16 consecutive rows (chunk = lane>>4), which is conflict-free from any base
// row with 2*((r>>2)&1) (found by exhaustive search over the ds_read_b128 lane groups).
template <bool M16>
__host__ __device__ inline int lds_swz(int r) {
  return M16 ? ((r >> 2) & 1) * 2 : (r >> 2) & 3;
}

// M16: 16-bit modes on v_mfma_f32_16x16x32 (the chip holds a higher clock on it than on 32x32x16
// at equal cycles per FLOP); weights of those layers are packed with the matching swizzle.
// NTERM = 2 (f16 activations, NQA_PREC_F32M's layers 1..6): the weights come as f16 (hi, lo) pairs, packed as TWO
// consecutive stages per kernel row -- [hi: 3 taps][lo: 3 taps] -- that contract the SAME activation rows into the
// same accumulators; nothing but the stage index -> (chunk, kernel row) map changes, and the epilogue multiplies by the
// layer's exact power-of-two 1/scale (the weights are packed times 2^k so that lo is a normal half).
template <typename P, int WAVES_N, int WAVES_M, int WN_T, int WM_T, int TW, bool M16, int NTERM = 1>
__global__ __launch_bounds__(64 * WAVES_N * WAVES_M) void conv3x3_igemm_kernel(
    const typename P::T *__restrict__ in, const char *__restrict__ wpk, const float *__restrict__ bias,
    typename P::T *__restrict__ out, int H, int W, int Cin, int Cout, int tiles_x, int out_split, float floor_v) {
#if defined(__HIP_DEVICE_COMPILE__)  // the LDS-DMA builtin exists in the device pass only
  typedef typename P::T T;
  typedef ConvGeom<WAVES_N, WAVES_M, WN_T, WM_T, TW> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A0][A1][W0][W1]

  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a provably wave-uniform (SGPR) value: every LDS-DMA destination is then
  // scalar arithmetic + s_mov m0 instead of a VGPR address + v_readfirstlane per DMA
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // 8-wave tiles: the two waves of a SIMD (w, w+4) run the same loop in the same phase; at different issue
  // priorities they stop competing instruction by instruction (measured +1.0-1.5 % on layers 5..9 in f16; none in
  // f32s and none on conv2_1's register-weights kernel, so it is not set there).  NQA_PRIO_SPLIT=0 builds the kernel without it.
#if !defined(NQA_PRIO_SPLIT)
#define NQA_PRIO_SPLIT 2
#endif
  if (NQA_PRIO_SPLIT && M16 && WAVES_N * WAVES_M == 8 && wave >= 4) __builtin_amdgcn_s_setprio(NQA_PRIO_SPLIT);
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so workgroup ids that
  // are equal mod 8 share an L2.  Give each of those classes a contiguous run of pixel tiles so
  // that neighbouring tiles' halo overlap is an L2 hit (speed only; any placement is correct).
  int tile_id = blockIdx.x;
  {
    const int nb = gridDim.x, qq = nb >> 3, rr = nb & 7, xcd = tile_id & 7, local = tile_id >> 3;
    tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
  }
  const int bx = tile_id % tiles_x, by = tile_id / tiles_x;
  const int n = blockIdx.y, ct = blockIdx.z;
  const int x0 = bx * TW, y0 = by * G::TH;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  const int nCC = Cin / P::KC;
  const int S = nCC * 3 * NTERM;

  // ---- LDS-DMA plan: halo tile.  Buffer loads with the image as the buffer: an out-of-image
  // halo pixel gets an out-of-range offset, so the DMA transfers nothing for it (or zeros);
  // those LDS cells are zeroed once below and stay zero -- that is the conv's zero padding.
  const unsigned kOOB = 0x80000000u;
  unsigned a_goff[G::A_ROUNDS];
#pragma unroll
  for (int r = 0; r < G::A_ROUNDS; ++r) {
    const int i = r * G::THREADS + tid;
    const int q = i >> 2, c = (i & 3) ^ lds_swz<M16>(q);
    const int hy = q / G::HW_, hx = q - hy * G::HW_;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = i < G::A_ITEMS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    a_goff[r] = ok ? (unsigned)(((gy * W + gx) * Cin + c * P::CPC) * (int)sizeof(T)) : kOOB;
    if (!ok && i < G::A_ITEMS_PAD) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4 *>(smem + i * 16) = z;
      *reinterpret_cast<u32x4 *>(smem + G::A_BYTES + i * 16) = z;
    }
  }
  const unsigned img_bytes = (unsigned)H * (unsigned)W * (unsigned)Cin * (unsigned)sizeof(T);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T *>(in + (size_t)n * H * W * Cin), 0, img_bytes, 0x00020000);
  // ---- LDS-DMA plan: weight rows (sub-slab j, item idx) ----
  const unsigned sub_stride = (unsigned)nCC * 9u * 64u * 64u * (unsigned)NTERM;  // bytes between 64-channel sub-slabs
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(wpk + (size_t)ct * G::NSUB * sub_stride), 0, G::NSUB * sub_stride, 0x00020000);
  unsigned w_goff[G::W_ROUNDS];
#pragma unroll
  for (int r = 0; r < G::W_ROUNDS; ++r) {
    const int i = r * G::THREADS + tid;
    w_goff[r] = (unsigned)(i / G::SUB_STAGE_ITEMS) * sub_stride + (unsigned)(i % G::SUB_STAGE_ITEMS) * 16u;
  }
  const int wave_base = wave * 64 * 16;  // this wave's 1 KB slot inside a round

  // hipcc does not count buffer-to-LDS DMA in its s_waitcnt bookkeeping (and with
  // global_load_lds it degrades every LDS wait in the loop to lgkmcnt(0)), so the DMA is
  // retired by hand: dma_wait() before the barrier that publishes a stage.
  auto issue = [&](int s) {
#ifndef NQA_ABLATE_NO_DMA
    const int cc = s / (3 * NTERM);
    char *wdst = smem + 2 * G::A_BYTES + (s & 1) * G::W_BYTES + wave_base;
#pragma unroll
    for (int r = 0; r < G::W_ROUNDS; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t *)(wdst + r * G::THREADS * 16), 16, w_goff[r],
                                               s * (G::SUB_STAGE_ITEMS * 16), 0, 0);
    if (s - cc * (3 * NTERM) == 0) {
      char *adst = smem + (cc & 1) * G::A_BYTES + wave_base;
#pragma unroll
      for (int r = 0; r < G::A_ROUNDS; ++r) {
        if (r * G::THREADS + wave * 64 < G::A_ITEMS_PAD)  // wave-uniform
          __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void_t *)(adst + r * G::THREADS * 16), 16, a_goff[r],
                                                   cc * 64, 0, 0);
      }
    }
#endif
  };
  auto dma_wait = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- per-lane LDS read addresses ----
  int w_base[WN_T], w_sw[WN_T];  // weight rows (MFMA rows = output channels)
#pragma unroll
  for (int i = 0; i < WN_T; ++i) {
    const int cib = (wn * WN_T + i) * 32 + l31;
    const int r64 = cib & 63;
    w_base[i] = 2 * G::A_BYTES + (cib >> 6) * (G::SUB_STAGE_ITEMS * 16) + r64 * 64;
    w_sw[i] = (r64 >> 2) & 3;
  }
  int q0[WM_T];  // halo index of this lane's output pixel at tap (0,0)
#pragma unroll
  for (int j = 0; j < WM_T; ++j) {
    const int m = (wm * WM_T + j) * 32 + l31;
    const int ty = m / TW, tx = m - ty * TW;
    q0[j] = ty * G::HW_ + tx;
  }

  f32x16 acc[WN_T][WM_T];
#pragma unroll
  for (int i = 0; i < WN_T; ++i)
#pragma unroll
    for (int j = 0; j < WM_T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // M16: the wave's WN_T x WM_T tiles of 32x32 as 2WN_T x 2WM_T tiles of 16x16 (4 accumulators each);
  // lane = (l15: row of the A fragment / column of the B fragment, c4: 16-byte k-chunk of the row)
  const int l15 = lane & 15, c4 = lane >> 4;
  int w16_addr[2 * WN_T], q16[2 * WM_T];
  f32x4 acc16[2 * WN_T][2 * WM_T];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 2 * WN_T; ++i) {
      const int cib = wn * WN_T * 32 + i * 16 + l15;
      const int r64 = cib & 63;
      w16_addr[i] = 2 * G::A_BYTES + (cib >> 6) * (G::SUB_STAGE_ITEMS * 16) + r64 * 64 + ((c4 ^ lds_swz<true>(r64)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2 * WM_T; ++j) {
      const int m = wm * WM_T * 32 + j * 16 + l15;
      const int ty = m / TW, tx = m - ty * TW;
      q16[j] = ty * G::HW_ + tx;
    }
#pragma unroll
    for (int i = 0; i < 2 * WN_T; ++i)
#pragma unroll
      for (int j = 0; j < 2 * WM_T; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

#ifdef NQA_STAMPS
  unsigned long long seg_sum[4] = {0, 0, 0, 0};
#endif
  issue(0);
  for (int s = 0; s < S; ++s) {
    const int cc = s / (3 * NTERM), ky = (s - cc * (3 * NTERM)) / NTERM;
    NQA_STAMP(t0);
    dma_wait();       // this wave's DMA for stage s has landed
    NQA_STAMP(t1);
    __syncthreads();  // ... and everyone's; every wave has also finished reading stage s-1
    NQA_STAMP(t2);
    if (s + 1 < S) issue(s + 1);
    NQA_STAMP(t3);
    NQA_STAMP_ADD(0, t0, t1);
    NQA_STAMP_ADD(1, t1, t2);
    NQA_STAMP_ADD(2, t2, t3);
    const char *abuf = smem + (cc & 1) * G::A_BYTES;
    const char *wbuf = smem + (s & 1) * G::W_BYTES;
    // six k-steps per stage (3 taps x 2 chunk pairs); fragments of step t+1 are read from LDS
    // while the MFMAs of step t run (two register sets, static indices)
    auto load_frags = [&](int t, u32x4(&af)[WN_T], u32x4(&bf)[WM_T]) {
      const int kx = t >> 1, ch = 2 * (t & 1) + h;
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
        af[i] = *reinterpret_cast<const u32x4 *>(wbuf + w_base[i] + kx * 4096 + ((ch ^ w_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < WM_T; ++j) {
        const int q = q0[j] + ky * G::HW_ + kx;
        bf[j] = *reinterpret_cast<const u32x4 *>(abuf + q * 64 + ((ch ^ ((q >> 2) & 3)) << 4));
      }
    };
    auto mma_all = [&](const u32x4(&af)[WN_T], const u32x4(&bf)[WM_T]) {
#ifndef NQA_ABLATE_NO_MFMA
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
#pragma unroll
        for (int j = 0; j < WM_T; ++j) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
#else
#pragma unroll
      for (int i = 0; i < WN_T; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
      for (int j = 0; j < WM_T; ++j) asm volatile("" ::"v"(bf[j]));
#endif
    };
    if constexpr (M16) {
      // three k-steps per stage (one tap = the whole 64-byte row = K 32), each in two halves of the
      // wave's channel rows: step t = (kx, half) reads WN_T weight fragments, and at half 0 the
      // 2*WM_T pixel fragments of the tap, one step ahead of the MFMAs that use them.
      u32x4 af[2][WN_T], bf[2][2 * WM_T];
      auto load16 = [&](int t) {
        const int kx = t >> 1, half = t & 1;
#pragma unroll
        for (int i = 0; i < WN_T; ++i)
          af[t & 1][i] = *reinterpret_cast<const u32x4 *>(wbuf + w16_addr[half * WN_T + i] + kx * 4096);
        if (half == 0) {
#pragma unroll
          for (int j = 0; j < 2 * WM_T; ++j) {
            const int q = q16[j] + ky * G::HW_ + kx;
            bf[kx & 1][j] = *reinterpret_cast<const u32x4 *>(abuf + q * 64 + ((c4 ^ lds_swz<true>(q)) << 4));
          }
        }
      };
      auto mma16 = [&](int t) {
        const int kx = t >> 1, half = t & 1;
#pragma unroll
        for (int i = 0; i < WN_T; ++i)
#pragma unroll
          for (int j = 0; j < 2 * WM_T; ++j) {
            f32x4 &c = acc16[half * WN_T + i][j];
            if constexpr (P::ID == NQA_PREC_BF16)
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[t & 1][i]),
                                                          __builtin_bit_cast(bf16x8, bf[kx & 1][j]), c, 0, 0, 0);
            else
              c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[t & 1][i]),
                                                         __builtin_bit_cast(f16x8, bf[kx & 1][j]), c, 0, 0, 0);
          }
      };
      load16(0);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t + 1 < 6) load16(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma16(t);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (P::SPLIT) {
      // split-f16 products: weight rows and pixel rows are both 16 channels as
      // [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] halves.  One k-step per tap: lane half h contracts
      // channels 8h..8h+7 (chunk h = hi, chunk 2+h = lo), three MFMAs per tile, taken term by term
      // so consecutive MFMAs never hit the same accumulator.
      auto load_split = [&](int kx, u32x4(&ah)[WN_T], u32x4(&al)[WN_T], u32x4(&bh)[WM_T], u32x4(&bl)[WM_T]) {
#pragma unroll
        for (int i = 0; i < WN_T; ++i) {
          const char *row = wbuf + w_base[i] + kx * 4096;
          ah[i] = *reinterpret_cast<const u32x4 *>(row + ((h ^ w_sw[i]) << 4));
          al[i] = *reinterpret_cast<const u32x4 *>(row + (((2 + h) ^ w_sw[i]) << 4));
        }
#pragma unroll
        for (int j = 0; j < WM_T; ++j) {
          const int q = q0[j] + ky * G::HW_ + kx, sw = (q >> 2) & 3;
          bh[j] = *reinterpret_cast<const u32x4 *>(abuf + q * 64 + ((h ^ sw) << 4));
          bl[j] = *reinterpret_cast<const u32x4 *>(abuf + q * 64 + (((2 + h) ^ sw) << 4));
        }
      };
      auto mma_split = [&](const u32x4(&ah)[WN_T], const u32x4(&al)[WN_T], const u32x4(&bh)[WM_T],
                           const u32x4(&bl)[WM_T]) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int i = 0; i < WN_T; ++i)
#pragma unroll
            for (int j = 0; j < WM_T; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                  __builtin_bit_cast(f16x8, term == 0 ? al[i] : ah[i]),
                  __builtin_bit_cast(f16x8, term == 1 ? bl[j] : bh[j]), acc[i][j], 0, 0, 0);
      };
      u32x4 ahA[WN_T], alA[WN_T], bhA[WM_T], blA[WM_T], ahB[WN_T], alB[WN_T], bhB[WM_T], blB[WM_T];
      load_split(0, ahA, alA, bhA, blA);
      load_split(1, ahB, alB, bhB, blB);
      __builtin_amdgcn_sched_barrier(0);
      mma_split(ahA, alA, bhA, blA);
      __builtin_amdgcn_sched_barrier(0);
      load_split(2, ahA, alA, bhA, blA);
      __builtin_amdgcn_sched_barrier(0);
      mma_split(ahB, alB, bhB, blB);
      __builtin_amdgcn_sched_barrier(0);
      mma_split(ahA, alA, bhA, blA);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    // sched_barrier(0) pins "reads of the next step, then MFMAs of this step": left alone,
    // hipcc sinks every ds_read to just before its MFMA and waits lgkmcnt(0) each time
    u32x4 afA[WN_T], bfA[WM_T], afB[WN_T], bfB[WM_T];
    load_frags(0, afA, bfA);
#pragma unroll
    for (int t = 0; t < 6; t += 2) {
      load_frags(t + 1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < 6) load_frags(t + 2, afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    NQA_STAMP(t4);
    NQA_STAMP_ADD(3, t3, t4);
  }
#ifdef NQA_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&g_stamps[i], seg_sum[i]);
    atomicAdd(&g_stamps[4], (unsigned long long)S);
  }
#endif

  // ---- epilogue: bias + ReLU, staged through LDS so the tile leaves as whole pixel records ----
  // acc[i][j][r]: channel = 8*(r>>2) + 4*h + (r&3) of row tile i, pixel = lane&31 of column tile j:
  // a lane owns 4 consecutive channels of one pixel, i.e. 8- or 16-byte pieces a whole record
  // (Cout * sizeof(T) bytes) apart -- stored directly, every lane would write its own cache line.
  // Instead each column tile j goes through LDS (the stage buffers are free now) as
  // [pixel][this block's RB = BN*sizeof(T) record bytes], 16-byte chunks XOR-swizzled by the row
  // so both sides are conflict-free, and leaves as 16 bytes per lane with consecutive lanes on
  // consecutive chunks of one record: full-line writes.
  constexpr int RB = G::BN * (int)sizeof(T), NCH = RB / 16, SWZ = NCH >= 16 ? 15 : NCH - 1;
  constexpr int ROWS = WAVES_M * 32;
  static_assert(ROWS * RB <= G::LDS_BYTES, "epilogue tile does not fit the stage buffers");
  char *const obase = reinterpret_cast<char *>(out) + (size_t)ct * RB;
  const size_t rec = (size_t)Cout * sizeof(T);
  // one lane's piece: 4 consecutive channels (first = cl, block-local) of the pixel staged in `row`
  // f32s: the weights were packed times a power of two (nqa_pack_vgg_weights); 1/scale sits behind the bias
  const float winv = (P::SPLIT || NTERM == 2) ? bias[Cout] : 1.f;
  // The wave's bias vectors are fetched HERE, all at once, and land under the barrier below.  Fetched inside
  // stage_piece they came one at a time, each waited for (global_load -> s_waitcnt vmcnt(0) -> ten VALU -> the next):
  // sixteen exposed cache round trips per tile on the one-block-per-CU tiles, where nothing else runs beside an epilogue.
  constexpr int NBQ = M16 ? 2 * WN_T : 4 * WN_T;
  f32x4 bq[NBQ];
#pragma unroll
  for (int q = 0; q < NBQ; ++q) {
    const int cl = M16 ? wn * WN_T * 32 + q * 16 + 4 * c4 : (wn * WN_T + (q >> 2)) * 32 + 8 * (q & 3) + 4 * h;
    bq[q] = *reinterpret_cast<const f32x4 *>(bias + ct * G::BN + cl);
  }
  auto stage_piece = [&](int row, int cl, const f32x4 &b4_pre, float a0, float a1, float a2, float a3) {
    char *const rbase = smem + row * RB;
    const int sw = row & SWZ;
#ifdef NQA_EPI_BIAS_INLINE  // A/B build: the round-2 form (one bias load per piece, waited for in place)
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + ct * G::BN + cl);
    (void)b4_pre;
#else
    const f32x4 &b4 = b4_pre;
#endif
    if constexpr (P::SPLIT || NTERM == 2) {  // exact: winv is a power of two
      a0 *= winv;
      a1 *= winv;
      a2 *= winv;
      a3 *= winv;
    }
    // floor_v = 0: the layer's ReLU; -inf: none (the data-gradient convolutions of nqa_backward.hip)
    const float v[4] = {fmaxf(a0 + b4[0], floor_v), fmaxf(a1 + b4[1], floor_v), fmaxf(a2 + b4[2], floor_v),
                        fmaxf(a3 + b4[3], floor_v)};
    if constexpr (sizeof(T) == 4) {
      if (P::SPLIT && out_split) {  // feeds another conv: split16 record (wave-uniform branch)
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        const h4 lo = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]),
                       (_Float16)(v[2] - (float)hi[2]), (_Float16)(v[3] - (float)hi[3])};
        const int off = (cl >> 4) * 64 + ((cl >> 3) & 1) * 16 + (cl & 7) * 2;
        *reinterpret_cast<h4 *>(rbase + ((((off >> 4)) ^ sw) << 4) + (off & 15)) = hi;
        *reinterpret_cast<h4 *>(rbase + ((((off >> 4) + 2) ^ sw) << 4) + (off & 15)) = lo;
      } else {
        const f32x4 s4 = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(rbase + (((cl >> 2) ^ sw) << 4)) = s4;
      }
    } else {
      typedef __attribute__((ext_vector_type(4))) T t4;
      const t4 s4 = {P::from_f(v[0]), P::from_f(v[1]), P::from_f(v[2]), P::from_f(v[3])};
      *reinterpret_cast<t4 *>(rbase + (((cl >> 3) ^ sw) << 4) + (cl & 4) * 2) = s4;
    }
  };
#pragma unroll
  for (int j = 0; j < WM_T; ++j) {
    __syncthreads();  // the last stage's LDS reads (or the previous pass's copy-out) are done
    if constexpr (M16) {
      // acc16[i][jj][e]: channel = 16*i + 4*c4 + e of the wave's rows, pixel = l15 of 16-pixel tile jj
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 2 * WN_T; ++i) {
          const f32x4 &c = acc16[i][2 * j + jj];
          stage_piece(wm * 32 + jj * 16 + l15, wn * WN_T * 32 + i * 16 + 4 * c4, bq[i], c[0], c[1], c[2], c[3]);
        }
    } else {
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          stage_piece(wm * 32 + l31, (wn * WN_T + i) * 32 + 8 * g + 4 * h, bq[M16 ? 0 : 4 * i + g], acc[i][j][4 * g],
                      acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    }
    __syncthreads();
#pragma unroll 2
    for (int idx = tid; idx < ROWS * NCH; idx += G::THREADS) {
      const int row = idx / NCH, k = idx - row * NCH;
      const int m = ((row >> 5) * WM_T + j) * 32 + (row & 31);
      const int ty = m / TW, tx = m - ty * TW;
      const int gy = y0 + ty, gx = x0 + tx;
      if (gy < H && gx < W)
        *reinterpret_cast<u32x4 *>(obase + ((size_t)(n * H + gy) * W + gx) * rec + k * 16) =
            *reinterpret_cast<const u32x4 *>(smem + row * RB + ((k ^ (row & SWZ)) << 4));
    }
  }
#endif
}

// ---------------------------------------------------------------------------------
// conv3x3 of a 64-input-channel layer with the WEIGHTS IN REGISTERS (16-bit modes; conv2_1 = layer 2).
// ---------------------------------------------------------------------------------
// With Cin = 64 the contraction is only 18 k-steps of 32 and the implicit-GEMM kernel above spends most of a
// stage behind its barrier (6 stages per tile, each shorter than an LDS-DMA round trip; measured 0.34 of the
// MFMA peak).  Here nothing is staged per k-step at all:
//   * a wave owns 32 output channels and keeps their 32 x 576 weights as 2 x 18 MFMA A fragments in 144 VGPRs
//     for the life of a persistent block (one block of 8 waves per CU: NCG = Cout/32 channel groups times
//     8/NCG pixel parts of an 8 x 32-pixel tile);
//   * the only thing that moves is the activation: the tile's 10 x 34 halo patch (both 32-channel chunks,
//     42.5 KB) arrives by LDS-DMA into a three-slot ring two tiles ahead of its use; ONE barrier per tile
//     publishes it.  Out-of-image halo cells are zeroed explicitly per tile (the conv's zero padding);
//   * per k-step a wave reads 4 pixel fragments (16 px x 32 ch, one ds_read_b128 each) and feeds each to its
//     two weight tiles: 0.5 LDS reads per v_mfma_f32_16x16x32, half of what the LDS can deliver;
//   * bias + ReLU in registers, results leave as 8-byte buffer stores (4 consecutive channels of a pixel);
//     out-of-image pixels get an out-of-range offset, so every wave issues the same number of stores and of
//     DMA pieces per tile and the ring is retired with ONE counted vmcnt per tile (stores and DMA count
//     together, in issue order).
// Tiles are dealt XCD-aware: the workgroups that share an L2 walk a contiguous range of tiles.
// NTERM = 2 (NQA_PREC_F32M): a wave owns 16 output channels and its two fragment sets are the f16 (hi, lo) parts of
// THEIR weights (packed times the layer's power-of-two scale); both parts contract the same pixel fragments into
// separate accumulators that are added, descaled and biased in the epilogue.  Same registers, same MFMAs and LDS
// reads per tile as the one-term form; the block covers its COUT channels with twice the channel groups.
template <typename P, int NCG, int NTERM = 1>
__global__ __launch_bounds__(512) void conv3x3_regw_kernel(const typename P::T *__restrict__ in,
                                                           const char *__restrict__ wreg,
                                                           const float *__restrict__ bias,
                                                           typename P::T *__restrict__ out, int H, int W, int tiles_x,
                                                           int tiles_y, int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename P::T T;
  constexpr int COUT = 32 * NCG, TH = 8, TW = 32, HWD = TW + 2, NQ = (TH + 2) * HWD;  // 340 halo pixels
  constexpr int CW = 32 / NTERM, NG = COUT / CW;   // channels per wave, channel groups per block
  constexpr int CH_ITEMS = 1536, CH_BYTES = CH_ITEMS * 16, SLOT = 2 * CH_BYTES;       // 3 DMA rounds per chunk
  constexpr int NPIECE = 6;                        // DMA pieces per wave and tile
  constexpr int RW = NG;                           // tile rows per wave (8 rows over 8/NG pixel parts)
  constexpr int GPP = 2;                           // 16-pixel groups per pass (one 32-pixel tile row)
  constexpr int NPASS = 2 * RW / GPP;              // passes per wave and tile
  constexpr int NSTORE = (2 / NTERM) * GPP * NPASS;  // buffer stores per wave and tile
  static_assert((NCG == 2 || NCG == 4) && (NTERM == 1 || NTERM == 2) && NG <= 8, "64 or 128 output channels");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // three halo slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = wave % NG, ph = wave / NG;

  // ---- this block's tiles: XCD x owns tiles [T*x/8, T*(x+1)/8), dealt round-robin to its blocks ----
  const int nblk = gridDim.x, nx = nblk < 8 ? nblk : 8;  // (a grid of fewer than 8 blocks has fewer classes)
  const int xcd = blockIdx.x % nx, jb = blockIdx.x / nx;
  const int blk_per_xcd = (nblk - xcd + nx - 1) / nx;  // blocks with id = xcd (mod nx)
  const int t_lo = (int)((long)total_tiles * xcd / nx), t_hi = (int)((long)total_tiles * (xcd + 1) / nx);
  const int my_tiles = t_lo + jb < t_hi ? (t_hi - t_lo - jb - 1) / blk_per_xcd + 1 : 0;
  if (my_tiles == 0) return;  // (block-uniform)
  auto tile_coords = [&](int it, int &n, int &x0, int &y0) {
    const int t = t_lo + jb + it * blk_per_xcd;
    n = t / (tiles_x * tiles_y);
    const int t2 = t - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * TW;
    y0 = by * TH;
  };

  // ---- weights: 2 tiles of 16 channels x 18 k-steps, straight into registers ----
  u32x4 wf[2][18];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cg * 2 + i) * 18 + ks) * 64 + lane) * 16);
  float bia[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) bia[i][e] = bias[NTERM == 2 ? cg * 16 + 4 * c4 + e : cg * 32 + i * 16 + 4 * c4 + e];
  const float winv = NTERM == 2 ? bias[COUT] : 1.f;  // 1 / the two-term weights' power-of-two scale

  // ---- halo DMA plan: item j (16 B) of a chunk = quarter (j&3)^swz(q) of halo pixel q = j>>2; rounds r and r+3
  // are the two chunks of the same (pixel, quarter) ----
  int p_hy[3], p_hx[3], p_c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int j = r * 512 + tid, q = j >> 2;
    p_hy[r] = q < NQ ? q / HWD : -100000;  // items past the patch: never inside any image
    p_hx[r] = q - (q / HWD) * HWD;
    // the chunk swizzle goes by the pixel's COLUMN in the patch, not by its linear index: equally conflict-free
    // (a row of the patch only shifts the phase of the pattern), and a tap's LDS address is then one per-lane
    // constant per kx plus compile-time offsets -- no address arithmetic in the k loop (see tap_base below)
    p_c[r] = (j & 3) ^ lds_swz<true>(p_hx[r]);
  }
  // per-lane byte offset of this lane's fragment of patch column l15 + kx (any patch row, 16-pixel group 0)
  int tap_base[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) tap_base[kx] = (l15 + kx) * 64 + ((c4 ^ lds_swz<true>(l15 + kx)) << 4);
  const unsigned kOOB = 0x80000000u;
  const unsigned img_in_bytes = (unsigned)H * (unsigned)W * 64u * (unsigned)sizeof(T);
  const unsigned img_out_bytes = (unsigned)H * (unsigned)W * (unsigned)COUT * (unsigned)sizeof(T);
  auto issue_halo = [&](int it, int slot_idx) {
    char *slot = smem + slot_idx * SLOT;
    const bool real = it < my_tiles;  // past the last tile: same number of pieces, all out of range
    int n = 0, x0 = 0, y0 = 0;
    if (real) tile_coords(it, n, x0, y0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(in + (size_t)n * H * W * 64), 0, img_in_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int gy = y0 - 1 + p_hy[r], gx = x0 - 1 + p_hx[r];
      const bool ok = real && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((gy * W + gx) * 64 + p_c[r] * 8) * (int)sizeof(T)) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(slot + r * 8192 + wave * 1024), 16, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(slot + CH_BYTES + r * 8192 + wave * 1024), 16,
                                               off, 64, 0, 0);
      if (real && !ok && p_hy[r] >= 0) {  // zero padding (the DMA transfers nothing for an out-of-range lane)
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4 *>(slot + (r * 512 + tid) * 16) = z;
        *reinterpret_cast<u32x4 *>(slot + CH_BYTES + (r * 512 + tid) * 16) = z;
      }
    }
  };

  issue_halo(0, 0);
  issue_halo(1, 1);
  for (int it = 0; it < my_tiles; ++it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    // the halo of tile `it` has landed: in issue order the only younger operations are the previous tile's
    // stores and the next tile's 6 pieces (always issued, see issue_halo)
    if (it == 0)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPIECE) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPIECE + NSTORE) : "memory");
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)n * H * W * COUT, 0, img_out_bytes, 0x00020000);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      // this pass's tile row: three address registers (one per kx), every tap / chunk / pixel group an immediate offset
      int rowb[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) rowb[kx] = tap_base[kx] + (it % 3) * SLOT + (ph * RW + pass) * (HWD * 64);
      // opaque per pass: otherwise hipcc hoists the fragment addresses of the unrolled k loop out of the
      // persistent tile loop and spills the weight registers
      asm volatile("" : "+v"(rowb[0]), "+v"(rowb[1]), "+v"(rowb[2]));
      f32x4 acc[2][GPP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < GPP; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // pixel fragments of k-step ks+1 are read while the MFMAs of ks run (two register sets; a third set, two steps
      // of lead, measured 10 % SLOWER on conv2_1)
      u32x4 bf[2][GPP];
      auto load_b = [&](int ks, u32x4(&b)[GPP]) {
        const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
        for (int g = 0; g < GPP; ++g)
          b[g] = *reinterpret_cast<const u32x4 *>(smem + rowb[kx] + (cc * CH_BYTES + ky * (HWD * 64) + g * 1024));
      };
      load_b(0, bf[0]);
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + 1 < 18) load_b(ks + 1, bf[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GPP; ++g)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (P::ID == NQA_PREC_BF16)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bf[ks & 1][g]), acc[i][g], 0, 0, 0);
            else
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                                 __builtin_bit_cast(f16x8, bf[ks & 1][g]), acc[i][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      // bias + ReLU; lane = (pixel l15 of the group, channels 4*c4.. of each 16-channel tile): 8-byte stores
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const int gy = y0 + ph * RW + pass, gx = x0 + g * 16 + l15;
        const bool inside = gy < H && gx < W;
        typedef __attribute__((ext_vector_type(4))) T t4;
        if constexpr (NTERM == 2) {
          t4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf((acc[0][g][e] + acc[1][g][e]) * winv + bia[0][e], 0.f));
          const unsigned off = inside ? (unsigned)(((gy * W + gx) * COUT + cg * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            t4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf(acc[i][g][e] + bia[i][e], 0.f));
            const unsigned off =
                inside ? (unsigned)(((gy * W + gx) * COUT + cg * 32 + i * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
          }
        }
      }
    }
    // slot (it+2)%3 was last read during tile it-1, which every wave left before this tile's barrier
    issue_halo(it + 2, (it + 2) % 3);
  }
#endif
}

// ---------------------------------------------------------------------------------
// The same for the 128-input-channel layers (conv2_2, conv3_1): ONE wave per SIMD, the whole register file.
// ---------------------------------------------------------------------------------
// 32 output channels x 1152 weights are 288 VGPRs -- too many for two waves per SIMD, so a block is 4 waves
// (launch bound 1 wave per SIMD, up to 512 registers each): wave = one 32-channel group of the block's 128 output
// channels, ALL 128 pixels of a 4 x 32 tile (4 passes of one tile row = 2 pixel groups).  The halo patch (6 x 34
// pixels x 4 chunks = 51 KB, 16 one-KB DMA pieces per wave) alternates between two LDS slots, requested one tile
// ahead; one barrier and one counted vmcnt per tile as in conv3x3_regw_kernel.  Every pixel fragment is read by
// all four waves (0.5 LDS reads per MFMA again); a tap address serves its four chunks through immediate offsets.
// Layers with 256 output channels run two such blocks per pixel tile (adjacent tile ids -> the halo hits L2).
// NTERM = 2 (NQA_PREC_F32M): a wave's two fragment sets are the (hi, lo) parts of 16 channels' weights, a block covers
// 64 output channels (Cout/64 blocks per pixel tile); see conv3x3_regw_kernel.
template <typename P, int PF, int NTERM = 1>
__global__ __launch_bounds__(256, 1) void conv3x3_regw128_kernel(const typename P::T *__restrict__ in,
                                                                const char *__restrict__ wreg,
                                                                const float *__restrict__ bias,
                                                                typename P::T *__restrict__ out, int H, int W, int Cout,
                                                                int tiles_x, int tiles_y, int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename P::T T;
  constexpr int CIN = 128, NCC = 4, NKS = NCC * 9, TH = 4, TW = 32, HWD = TW + 2, NQ = (TH + 2) * HWD;  // 204
  constexpr int CH_ITEMS = 1024, CH_BYTES = CH_ITEMS * 16, SLOT = NCC * CH_BYTES;  // 4 DMA rounds of 256 per chunk
  constexpr int GPP = 2, NPASS = TH, NSTORE = (2 / NTERM) * GPP * NPASS;  // (16 DMA pieces and 16 | 8 stores per wave and tile)
  constexpr int BC = 128 / NTERM;  // output channels per block
  extern __shared__ __attribute__((aligned(16))) char smem[];  // two halo slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = channel group of the block's 128 channels
  const int l15 = lane & 15, c4 = lane >> 4;
  const int nct = Cout / BC;

  const int nblk = gridDim.x, nx = nblk < 8 ? nblk : 8;
  const int xcd = blockIdx.x % nx, jb = blockIdx.x / nx;
  const int blk_per_xcd = (nblk - xcd + nx - 1) / nx;
  const int t_lo = (int)((long)total_tiles * xcd / nx), t_hi = (int)((long)total_tiles * (xcd + 1) / nx);
  const int my_tiles = t_lo + jb < t_hi ? (t_hi - t_lo - jb - 1) / blk_per_xcd + 1 : 0;
  if (my_tiles == 0) return;  // (block-uniform)
  auto tile_coords = [&](int it, int &n, int &x0, int &y0, int &ct) {
    const int t = t_lo + jb + it * blk_per_xcd;
    ct = t % nct;  // channel tile fastest: the blocks of one pixel tile are neighbours in time and place
    const int tp = t / nct;
    n = tp / (tiles_x * tiles_y);
    const int t2 = tp - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * TW;
    y0 = by * TH;
  };

  // ---- weights: 2 tiles of 16 channels x 36 k-steps; the channel tile of this block's first tile (re-loaded if a
  // later tile belongs to another channel tile: with nct = 2 consecutive tiles alternate, so blocks whose stride is
  // even keep theirs) ----
  u32x4 wf[2][NKS];
  float bia[2][4];
  int cur_ct = -1;
  auto load_weights = [&](int ct) {
    const int g = ct * 4 + wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)g * 2 + i) * NKS + ks) * 64 + lane) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) bia[i][e] = bias[NTERM == 2 ? g * 16 + 4 * c4 + e : g * 32 + i * 16 + 4 * c4 + e];
    cur_ct = ct;
  };
  const float winv = NTERM == 2 ? bias[Cout] : 1.f;

  // ---- halo DMA plan: 4 rounds of 256 items per chunk; item j = quarter (j&3)^swz(q) of halo pixel q = j>>2 ----
  int p_hy[4], p_hx[4], p_c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = r * 256 + tid, q = j >> 2;
    p_hy[r] = q < NQ ? q / HWD : -100000;
    p_hx[r] = q - (q / HWD) * HWD;
    p_c[r] = (j & 3) ^ lds_swz<true>(p_hx[r]);  // swizzled by the patch COLUMN (see conv3x3_regw_kernel)
  }
  int tap_base[3];  // per-lane byte offset of the fragment of patch column l15 + kx
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) tap_base[kx] = (l15 + kx) * 64 + ((c4 ^ lds_swz<true>(l15 + kx)) << 4);
  const unsigned kOOB = 0x80000000u;
  const unsigned img_in_bytes = (unsigned)H * (unsigned)W * (unsigned)CIN * (unsigned)sizeof(T);
  const unsigned img_out_bytes = (unsigned)H * (unsigned)W * (unsigned)Cout * (unsigned)sizeof(T);
  auto issue_halo = [&](int it, int slot_idx) {
    char *slot = smem + slot_idx * SLOT;
    const bool real = it < my_tiles;
    int n = 0, x0 = 0, y0 = 0, ct = 0;
    if (real) tile_coords(it, n, x0, y0, ct);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(in + (size_t)n * H * W * CIN), 0, img_in_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gy = y0 - 1 + p_hy[r], gx = x0 - 1 + p_hx[r];
      const bool ok = real && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((gy * W + gx) * CIN + p_c[r] * 8) * (int)sizeof(T)) : kOOB;
#pragma unroll
      for (int cc = 0; cc < NCC; ++cc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(slot + cc * CH_BYTES + r * 4096 + wave * 1024), 16,
                                                 off, cc * 64, 0, 0);
      if (real && !ok && p_hy[r] >= 0) {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) *reinterpret_cast<u32x4 *>(slot + cc * CH_BYTES + (r * 256 + tid) * 16) = z;
      }
    }
  };

  issue_halo(0, 0);
  for (int it = 0; it < my_tiles; ++it) {
    int n, x0, y0, ct;
    tile_coords(it, n, x0, y0, ct);
    if (ct != cur_ct) {  // (block-uniform; compiler-tracked loads)
      load_weights(ct);
      // Retire them HERE, with a wait the compiler can see (the builtin, not inline asm): left pending, the fragments
      // are "values loaded outside a store-only loop that uses them" at the pass loop below, and for that shape the
      // waitcnt pass drains vmcnt(0) in front of the loop -- on EVERY tile, right behind issue_halo(it + 1), i.e. the
      // next tile's halo was waited for before this tile's first MFMA (found in the ISA, round 3)
#ifndef NQA_NO_VISIBLE_WAIT  // (A/B build switch)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
#endif
    }
    // halo `it` has landed: the only younger operations are the previous tile's stores
    if (it == 0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
    issue_halo(it + 1, (it + 1) & 1);  // that slot was last read during tile it-1
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)n * H * W * Cout, 0, img_out_bytes, 0x00020000);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      int rowb[3];  // this pass's tile row: one address register per kx, everything else an immediate offset
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) rowb[kx] = tap_base[kx] + (it & 1) * SLOT + pass * (HWD * 64);
      asm volatile("" : "+v"(rowb[0]), "+v"(rowb[1]), "+v"(rowb[2]));
      f32x4 acc[2][GPP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < GPP; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 bf[PF + 1][GPP];  // pixel fragments, PF k-steps ahead of the MFMAs
      auto load_b = [&](int ks, u32x4(&b)[GPP]) {
        const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
        for (int g = 0; g < GPP; ++g)
          b[g] = *reinterpret_cast<const u32x4 *>(smem + rowb[kx] + (cc * CH_BYTES + ky * (HWD * 64) + g * 1024));
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
          for (int i = 0; i < 2; ++i) {
            if constexpr (P::ID == NQA_PREC_BF16)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bf[ks % (PF + 1)][g]), acc[i][g], 0, 0, 0);
            else
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                                 __builtin_bit_cast(f16x8, bf[ks % (PF + 1)][g]), acc[i][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const int gy = y0 + pass, gx = x0 + g * 16 + l15;
        const bool inside = gy < H && gx < W;
        typedef __attribute__((ext_vector_type(4))) T t4;
        if constexpr (NTERM == 2) {
          t4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf((acc[0][g][e] + acc[1][g][e]) * winv + bia[0][e], 0.f));
          const unsigned off =
              inside ? (unsigned)(((gy * W + gx) * Cout + ct * 64 + wave * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            t4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf(acc[i][g][e] + bia[i][e], 0.f));
            const unsigned off =
                inside ? (unsigned)(((gy * W + gx) * Cout + ct * 128 + wave * 32 + i * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
          }
        }
      }
    }
  }
#endif
}

// ---------------------------------------------------------------------------------
// Stage 1 (conv1_1 + conv1_2) on the same principle, 16-bit modes: the shipped form.
// ---------------------------------------------------------------------------------
// conv1_2 is conv3x3_regw_kernel with NCG = 2 (a wave = 32 of the 64 output channels x 2 rows of an 8 x 32 tile,
// its 2 x 18 weight fragments in registers), but its halo patch is not loaded: relu1_1 never exists in HBM.  Per
// tile the block
//   1. fetches the 12 x 36 raw pixels under the 10 x 34 halo patch (3 float planes -> one thread per pixel, loads
//      issued before the previous tile's conv1_2 so that they land under its MFMAs), normalises them
//      ((x-mean)/std, zero outside the image = conv1_1's padding) into an LDS patch of [row][col][4 halfs];
//   2. runs conv1_1 for the 340 halo pixels on the matrix cores: with the contraction ordered
//      k = 16*ky + 4*kx + c (kx and c padded to 4) a lane's 8-element B fragment is 16 contiguous bytes of that
//      patch, so one 16-pixel x 16-channel tile is two v_mfma_16x16x32 (kernel rows 0-1, then row 2 + zeros);
//      bias + ReLU, zero outside the image (conv1_2's padding), stored as halfs straight into the swizzled
//      [chunk][pixel][64 B] halo image conv1_2 reads (22 groups of 16 pixels over the 8 waves);
//   3. conv1_2 from that image, bias + ReLU, 8-byte stores.
// Two halo images alternate; two barriers per tile (patch visible, halo visible).  conv1_1 is 6 % of the FLOPs
// and costs the waves ~15 % of a tile here (it is not overlapped with conv1_2 of another wave group as in
// conv1_fused_kernel below), against conv1_2 running at 0.5 LDS reads per MFMA instead of 1.5.
// NTERM = 2 (NQA_PREC_F32M): both convolutions on two-term weights.  conv1_2 as conv3x3_regw_kernel<.., NTERM = 2> (a
// wave = 16 output channels x 4 tile rows, fragment sets = the hi / lo parts); conv1_1's fragments come as (hi, lo) pairs
// too (twice the MFMAs of a phase that is 6 % of the FLOPs), its accumulator starts at bias x scale and is descaled
// before the ReLU.  *w1inv_p = 1 / conv1_1's power-of-two weight scale (not read in the one-term form).
template <typename P, int NTERM = 1>
__global__ __launch_bounds__(512) void conv1_regw_kernel(const float *__restrict__ x, const float *__restrict__ y, int B,
                                                         const char *__restrict__ w1m, const float *__restrict__ bias1,
                                                         const char *__restrict__ wreg, const float *__restrict__ bias2,
                                                         typename P::T *__restrict__ out, int H, int W, int tiles_x,
                                                         int tiles_y, int total_tiles, const float *__restrict__ w1inv_p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename P::T T;
  typedef __attribute__((ext_vector_type(4))) T t4;
  constexpr int COUT = 64, TH = 8, TW = 32, HWD = TW + 2, NQ = (TH + 2) * HWD;  // 340 halo pixels
  // halo image [chunk][pixel][96 B]: a pixel's 32 channels (64 B) + 32 B of padding.  With a 96-byte pitch a
  // ds_read_b128 of 16 consecutive pixels at one 16-byte quarter is bank-conflict-free WITHOUT an XOR swizzle (the
  // image is written by ds_write here, not by 1-KB LDS-DMA pieces, so the pitch is free), and then a tap is a
  // compile-time byte offset from one per-group base register: no address arithmetic in the k loop.
  constexpr int PITCH = 96, CH_BYTES = NQ * PITCH, SLOT = 2 * CH_BYTES;  // (the last record either convolution touches: 339)
  constexpr int RAWP = 40, RAW_ROWS = 13, RAW_BYTES = RAW_ROWS * RAWP * 8;  // (+1 row read, with zero weights, by MFMA 1)
  constexpr int RAW_OFF = 2 * SLOT, W1_OFF = RAW_OFF + 2 * RAW_BYTES, B1_OFF = W1_OFF + NTERM * 4 * 2 * 64 * 16;  // two raw patches
  constexpr int STG_OFF = B1_OFF + 256;  // float staging of the raw pixels, [3 planes][512 threads] (filled by LDS-DMA)
  constexpr int CW = 32 / NTERM, NG = COUT / CW;  // channels per wave, channel groups (2 of 32 | 4 of 16)
  constexpr int RW = NG, GPP = 2, NPASS = 2 * RW / GPP;
  constexpr int NGRP = (NQ + 15) / 16;  // 22 groups of 16 halo pixels
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [halo 0][halo 1][raw 0][raw 1][conv1_1 fragments][bias1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = wave % NG, ph = wave / NG;
  const int HW = H * W;
  const float w1inv = NTERM == 2 ? w1inv_p[0] : 1.f;

  const int nblk = gridDim.x, nx = nblk < 8 ? nblk : 8;
  const int xcd = blockIdx.x % nx, jb = blockIdx.x / nx;
  const int blk_per_xcd = (nblk - xcd + nx - 1) / nx;
  const int t_lo = (int)((long)total_tiles * xcd / nx), t_hi = (int)((long)total_tiles * (xcd + 1) / nx);
  const int my_tiles = t_lo + jb < t_hi ? (t_hi - t_lo - jb - 1) / blk_per_xcd + 1 : 0;
  if (my_tiles == 0) return;  // (block-uniform)
  auto tile_coords = [&](int it, int &n, int &x0, int &y0) {
    const int t = t_lo + jb + it * blk_per_xcd;
    n = t / (tiles_x * tiles_y);
    const int t2 = t - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * TW;
    y0 = by * TH;
  };

  // ---- one-time: conv1_2 weights into registers; conv1_1 fragments, bias1 and a zeroed raw patch into LDS ----
  u32x4 wf[2][18];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cg * 2 + i) * 18 + ks) * 64 + lane) * 16);
  float bia[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) bia[i][e] = bias2[NTERM == 2 ? cg * 16 + 4 * c4 + e : cg * 32 + i * 16 + 4 * c4 + e];
  const float winv2 = NTERM == 2 ? bias2[COUT] : 1.f;
#pragma unroll
  for (int r = 0; r < NTERM; ++r)  // 8 KB = 512 x 16 B per term
    reinterpret_cast<u32x4 *>(smem + W1_OFF)[r * 512 + tid] = reinterpret_cast<const u32x4 *>(w1m)[r * 512 + tid];
  if (tid < 64) reinterpret_cast<float *>(smem + B1_OFF)[tid] = NTERM == 2 ? bias1[tid] / w1inv : bias1[tid];  // (exact)
  for (int i = tid; i < 2 * RAW_BYTES / 8; i += 512) reinterpret_cast<u32x2 *>(smem + RAW_OFF)[i] = (u32x2){0u, 0u};

  // ---- raw patch: thread t < 432 owns pixel (t / 36, t % 36) of the 12 x 36 patch ----
  // The three float planes of a pixel travel by LDS-DMA into a staging area ([plane][thread], one dword per lane) when
  // the tile starts and are normalised into the raw patch a phase later (raw_commit): nothing waits on them in between.
  // As plain loads they did: hipcc hoisted `pixel - mean` to right behind the loads, and with the consumer moved away
  // it still drained vmcnt(0) in front of the phase loop (a store-only loop that uses a value loaded outside it), so
  // every wave sat out an HBM round trip per tile -- 230 us of a 2 580 us launch at 1080p (timing build without the
  // loads: 2 350 us).  The DMA is retired by hand with a counted vmcnt, as everywhere else in this file.
  const int r_row = tid / 36, r_col = tid - r_row * 36;
  const bool r_mine = tid < 12 * 36;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  bool r_ok = false;
  auto raw_fetch = [&](int it) {
    r_ok = false;
#ifdef NQA_R_NO_RAW  // timing-only ablation: no raw pixel loads at all (results are wrong on purpose)
    return;
#endif
    if (it < my_tiles) {  // (block-uniform; every wave issues its three pieces, so the counted waits below hold)
      int n, x0, y0;
      tile_coords(it, n, x0, y0);
      const int gy = y0 - 2 + r_row, gx = x0 - 2 + r_col;
      r_ok = r_mine && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float *img = n < B ? x + (size_t)n * 3 * HW : y + (size_t)(n - B) * 3 * HW;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, 3u * (unsigned)HW * 4u, 0x00020000);
      const unsigned off = r_ok ? (unsigned)((gy * W + gx) * 4) : 0x80000000u;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(smem + STG_OFF + c * 2048 + wave * 256), 4, off,
                                                 c * HW * 4, 0, 0);
    }
  };
  // `younger`: VMEM operations this wave has issued since raw_fetch (they retire in order): the 8 stores of conv1_2
  // for the waves that run it before conv1_1, none otherwise
  static_assert((2 / NTERM) * GPP * NPASS == 8, "raw_commit's counted wait assumes 8 stores per wave and tile");
  auto raw_commit = [&](int buf, bool stores_since) {
    if (stores_since)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (r_mine) {
      t4 v;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float raw = *reinterpret_cast<const float *>(smem + STG_OFF + (c * 512 + tid) * 4);
        v[c] = P::from_f(r_ok ? (raw - mean[c]) / sd[c] : 0.f);
      }
      v[3] = P::from_f(0.f);
      *reinterpret_cast<t4 *>(smem + RAW_OFF + buf * RAW_BYTES + (r_row * RAWP + r_col) * 8) = v;
    }
  };
  // ---- conv1_1 of tile `it` (raw patch it&1) into halo image it&1: all waves, groups wave, wave+8, wave+16 in
  // flight together (three independent read -> MFMA -> write chains instead of one after the other) ----
  auto conv1_1_halo = [&](int it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    char *slot = smem + (it & 1) * SLOT;
    const char *rawp = smem + RAW_OFF + (it & 1) * RAW_BYTES;
    constexpr int NU = (NGRP + 7) / 8;  // 3
    f32x4 a1[NU][4];
    int qv[NU], hyv[NU], hxv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = (wave + 8 * u) * 16 + l15, qc = q < NQ ? q : NQ - 1;
      qv[u] = q;
      hyv[u] = qc / HWD;
      hxv[u] = qc - hyv[u] * HWD;
#pragma unroll
      for (int i = 0; i < 4; ++i)  // the accumulator starts at the bias
        a1[u][i] = *reinterpret_cast<const f32x4 *>(smem + B1_OFF + (16 * i + 4 * c4) * 4);
    }
    // a tile whose whole halo patch lies inside the image (all but the frame's border tiles) needs no padding test
    const bool interior = y0 >= 1 && y0 + TH + 1 <= H && x0 >= 1 && x0 + TW + 1 <= W;  // (block-uniform)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u32x4 bfr[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const char *rp = rawp + ((hyv[u] + 2 * m + (c4 >> 1)) * RAWP + hxv[u] + (c4 & 1) * 2) * 8;
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(rp), hi = *reinterpret_cast<const u32x2 *>(rp + 8);
        bfr[u] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int part = 0; part < NTERM; ++part)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 wfr =
              *reinterpret_cast<const u32x4 *>(smem + W1_OFF + ((((i * 2 + m) * NTERM + part) * 64 + lane) * 16));
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if (wave + 8 * u < NGRP) {  // (wave-uniform)
              if constexpr (P::ID == NQA_PREC_BF16)
                a1[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr), __builtin_bit_cast(bf16x8, bfr[u]), a1[u][i], 0, 0, 0);
              else
                a1[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wfr), __builtin_bit_cast(f16x8, bfr[u]), a1[u][i], 0, 0, 0);
            }
          }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = qv[u];
      const int gy = y0 - 1 + hyv[u], gx = x0 - 1 + hxv[u];
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      if (wave + 8 * u < NGRP && q < NQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // channels 16*i + 4*c4 .. +3: chunk i>>1, quarter 2*(i&1) + (c4>>1), half (c4&1)
          t4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf(NTERM == 2 ? a1[u][i][e] * w1inv : a1[u][i][e], 0.f));
          if (!interior && !inside) v = (t4){P::from_f(0.f), P::from_f(0.f), P::from_f(0.f), P::from_f(0.f)};
          *reinterpret_cast<t4 *>(slot + (i >> 1) * CH_BYTES + q * PITCH + ((2 * (i & 1) + (c4 >> 1)) << 4) +
                                  (c4 & 1) * 8) = v;
        }
      }
    }
  };

  const unsigned kOOB = 0x80000000u;
  const unsigned img_out_bytes = (unsigned)H * (unsigned)W * (unsigned)COUT * (unsigned)sizeof(T);
  // Pipeline per tile k: raw pixels fetched at tile k-2 (start) and committed to raw patch k&1 (end of k-2),
  // conv1_1 -> halo image k&1 at tile k-1, conv1_2 at tile k.  ONE barrier per tile: behind it halo k (written
  // during k-1) and raw patch k+1 (committed during k-1) are visible, and halo image (k+1)&1 / raw patch k&1,
  // last read during tile k-1, are free.  The two waves of a SIMD (w, w+4) take conv1_1 and conv1_2 in opposite
  // order, so the latency-bound conv1_1 of one runs beside the MFMA-bound conv1_2 of the other.
  __syncthreads();  // zeroed patches, conv1_1 fragments and bias are in LDS
  raw_fetch(0);
  raw_commit(0, false);
  raw_fetch(1);
  raw_commit(1, false);
  __syncthreads();
  conv1_1_halo(0);
  auto conv1_2_tile = [&](int it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    const char *slot = smem + (it & 1) * SLOT;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        out + (size_t)n * H * W * COUT, 0, img_out_bytes, 0x00020000);
    constexpr int NSET = 2;  // fragment register sets (three -- two k-steps of lead -- measured neutral to -1 % here)
    u32x4 bf[NSET][GPP];
    auto load_b = [&](const int (&q0)[GPP], int ks, u32x4(&b)[GPP]) {
      const int cc = ks / 9, t = ks - cc * 9, ky = t / 3, kx = t - ky * 3;
#pragma unroll
      for (int g = 0; g < GPP; ++g)
        b[g] = *reinterpret_cast<const u32x4 *>(slot + q0[g] + (cc * CH_BYTES + (ky * HWD + kx) * PITCH));
    };
    int q0[GPP];  // byte offset of this lane's fragment at tap (0,0), chunk 0, per 16-pixel group
#pragma unroll
    for (int g = 0; g < GPP; ++g) q0[g] = ((ph * RW) * HWD + g * 16 + l15) * PITCH + (c4 << 4);
    asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
      f32x4 acc[2][GPP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < GPP; ++g) acc[i][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // pixel fragments of k-step ks+1 are read while the MFMAs of ks run (two register sets; a third set, two steps
      // of lead, measured 10 % SLOWER on conv2_1)
      load_b(q0, 0, bf[0]);
      if (NSET == 3) load_b(q0, 1, bf[1]);
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + NSET - 1 < 18) load_b(q0, ks + NSET - 1, bf[(ks + NSET - 1) % NSET]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GPP; ++g)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (P::ID == NQA_PREC_BF16)
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][ks]),
                                                                  __builtin_bit_cast(bf16x8, bf[ks % NSET][g]), acc[i][g], 0, 0, 0);
            else
              acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i][ks]),
                                                                 __builtin_bit_cast(f16x8, bf[ks % NSET][g]), acc[i][g], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const int gy = y0 + ph * RW + pass, gx = x0 + g * 16 + l15;
        const bool inside = gy < H && gx < W;
        if constexpr (NTERM == 2) {
          t4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf((acc[0][g][e] + acc[1][g][e]) * winv2 + bia[0][e], 0.f));
          const unsigned off = inside ? (unsigned)(((gy * W + gx) * COUT + cg * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            t4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = P::from_f(fmaxf(acc[i][g][e] + bia[i][e], 0.f));
            const unsigned off =
                inside ? (unsigned)(((gy * W + gx) * COUT + cg * 32 + i * 16 + 4 * c4) * (int)sizeof(T)) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), orsrc, off, 0, 0);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < GPP; ++g) q0[g] += HWD * PITCH;  // next tile row
      asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
    }
  };
#ifdef NQA_STAMPS  // per wave: [0] barrier wait, [1] conv1_2 k-loops + epilogue, [2] conv1_1, [3] raw fetch + commit
  unsigned long long seg_sum[4] = {0, 0, 0, 0};
#endif
  for (int it = 0; it < my_tiles; ++it) {
    NQA_STAMP(s0);
    __syncthreads();
    NQA_STAMP(s1);
    NQA_STAMP_ADD(0, s0, s1);
    raw_fetch(it + 2);  // lands under this tile's MFMAs
    NQA_STAMP(s2);
    NQA_STAMP_ADD(3, s1, s2);
    const bool next = it + 1 < my_tiles;
#if defined(NQA_R_NO_P1)  // timing-only ablations (tools/gpu_fused_bench.py; results are wrong on purpose)
    conv1_2_tile(it);
    (void)next;
#elif defined(NQA_R_NO_P2)
    if (next) conv1_1_halo(it + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no stores in this ablation: the counted wait below would not hold)
#else
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {  // (a loop, so that each phase's code exists once)
      // conv1_1 is a latency chain (LDS read -> 8 MFMAs -> convert -> LDS write, three groups in flight): at a
      // higher issue priority its few instructions go out the moment they are ready and the partner wave's
      // conv1_2 fills every other slot (2.70 -> 2.59 ms at 1080p, 409 -> 386 us at 256 x 256; priority 3: 2.65)
      NQA_STAMP(h0);
      if ((half == 0) == (wave < 4)) {
        __builtin_amdgcn_s_setprio(0);
        conv1_2_tile(it);
        NQA_STAMP(h1);
        NQA_STAMP_ADD(1, h0, h1);
      } else if (next) {
        __builtin_amdgcn_s_setprio(2);
        conv1_1_halo(it + 1);
        NQA_STAMP(h1);
        NQA_STAMP_ADD(2, h0, h1);
      }
    }
#endif
    // tile it+2's patch (patch it&1 was last read by conv1_1 of tile `it`, during tile it-1).  The pixels were requested
    // a whole tile ago; every wave has issued exactly its 8 stores since (in either phase order), so vmcnt(8) retires
    // the pixels without waiting for a single store
    NQA_STAMP(s3);
    raw_commit(it & 1, true);
    NQA_STAMP(s4);
    NQA_STAMP_ADD(3, s3, s4);
  }
#ifdef NQA_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&g_stamps[i], seg_sum[i]);
    atomicAdd(&g_stamps[4], (unsigned long long)my_tiles);
    if (wave < 4) atomicAdd(&g_stamps[5], seg_sum[1]); else atomicAdd(&g_stamps[6], seg_sum[1]);
  }
#endif
#endif
}

// ---------------------------------------------------------------------------------
// Stage 1 in f32s (conv1_1 + conv1_2, float32-class products), fused on the same principle.
// ---------------------------------------------------------------------------------
// What conv1_regw_kernel<PrecF16, 2> is to the two-term modes, with the ACTIVATIONS in two parts as well: a product is
// a_hi*w_hi + a_hi*w_lo + a_lo*w_hi (three 16x16x32 MFMAs, the dropped a_lo*w_lo is 2^-22 relative), relu1_1 never
// exists in HBM (the unfused path writes it as 8.5 GB of split16 records per 16 images of 1080p and reads it back),
// and conv1_1 runs on the matrix cores instead of the VALU (conv1_1_kernel<PrecF32S>).
//   * 4 x 32 output tiles (an 8-row tile's halo image does not fit twice): the halo is 6 x 34 = 204 pixels; a pixel's
//     record in the halo image is [32 channels hi | 32 channels lo | 32 B pad] = 160 B per 32-channel chunk, a pitch at
//     which a ds_read_b128 of 16 consecutive pixels is bank-conflict-free; two images alternate (133 KB);
//   * the raw patch is 8 x 36 pixels, normalised in float and stored as a hi and a lo plane of [row][col][4 halfs];
//     conv1_1 = 3 terms x 2 MFMAs (kernel rows 0-1 | row 2) x 4 channel tiles per 16 halo pixels, accumulator started at
//     bias x scale, descaled, ReLU, zero outside the image, split into (hi, lo) halves and written into the halo image;
//   * conv1_2: a wave = 16 output channels x 2 tile rows x 32 columns, its (hi, lo) weight fragments in registers (144
//     VGPRs, the two-term blob format), three accumulators per 16-pixel group (hi*hi | lo_w*hi_a | hi_w*lo_a) summed in
//     the epilogue; float NHWC out (the tap the statistics and the pool read).
// Same pipeline as conv1_regw_kernel: one barrier per tile, the two waves of a SIMD take conv1_1 (of the next tile) and
// conv1_2 (of this one) in opposite order.  LDS: 2 x 65 280 (halo) + 2 x 5 760 (raw) + 16 384 (conv1_1 fragments) + 256
// (bias) + 3 840 (raw-pixel staging) + 256 (conv1_2's bias) = 162 816 of the 163 840 bytes.
__global__ __launch_bounds__(512) void conv1_regw_split_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                               int B, const char *__restrict__ w1m,
                                                               const float *__restrict__ bias1,
                                                               const char *__restrict__ wreg,
                                                               const float *__restrict__ bias2, float *__restrict__ out,
                                                               int H, int W, int tiles_x, int tiles_y, int total_tiles,
                                                               const float *__restrict__ w1inv_p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  constexpr int COUT = 64, TH = 4, TW = 32, HWD = TW + 2, NQ = (TH + 2) * HWD;  // 204 halo pixels
  constexpr int NGRP = (NQ + 15) / 16;                                          // 13 groups of 16
  constexpr int PITCH = 160, LO = 64, CH_BYTES = NQ * PITCH, SLOT = 2 * CH_BYTES;  // (last record touched: 203)
  constexpr int RAWP = 40, RAW_ROWS = 9, RAW_PLANE = RAW_ROWS * RAWP * 8, RAW_BYTES = 2 * RAW_PLANE;  // hi plane, lo plane
  constexpr int RAW_OFF = 2 * SLOT, W1_OFF = RAW_OFF + 2 * RAW_BYTES, B1_OFF = W1_OFF + 2 * 4 * 2 * 64 * 16;
  constexpr int STG_OFF = B1_OFF + 256, STG_T = 320;  // float staging of the raw pixels, [3 planes][waves 0..4] (LDS-DMA)
  constexpr int B2_OFF = STG_OFF + 3 * STG_T * 4;     // conv1_2's bias (the epilogue reads it: five registers fewer)
  constexpr int NG = 4, RW = 2, GPP = 2, NPASS = RW;  // 4 channel groups of 16; a wave: 2 rows x 2 groups of 16 columns
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [halo 0][halo 1][raw 0][raw 1][conv1_1 fragments][bias1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = wave % NG, ph = wave / NG;
  const int HW = H * W;
  const float w1inv = w1inv_p[0];

  const int nblk = gridDim.x, nx = nblk < 8 ? nblk : 8;
  const int xcd = blockIdx.x % nx, jb = blockIdx.x / nx;
  const int blk_per_xcd = (nblk - xcd + nx - 1) / nx;
  const int t_lo = (int)((long)total_tiles * xcd / nx), t_hi = (int)((long)total_tiles * (xcd + 1) / nx);
  const int my_tiles = t_lo + jb < t_hi ? (t_hi - t_lo - jb - 1) / blk_per_xcd + 1 : 0;
  if (my_tiles == 0) return;  // (block-uniform)
  auto tile_coords = [&](int it, int &n, int &x0, int &y0) {
    const int t = t_lo + jb + it * blk_per_xcd;
    n = t / (tiles_x * tiles_y);
    const int t2 = t - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * TW;
    y0 = by * TH;
  };

  // ---- one-time: conv1_2 (hi, lo) fragments into registers; conv1_1 fragments, bias1 and zeroed raw patches into LDS ----
  u32x4 wf[2][18];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cg * 2 + i) * 18 + ks) * 64 + lane) * 16);
  const float winv2 = bias2[COUT];
#pragma unroll
  for (int r = 0; r < 2; ++r)  // 8 KB = 512 x 16 B per term
    reinterpret_cast<u32x4 *>(smem + W1_OFF)[r * 512 + tid] = reinterpret_cast<const u32x4 *>(w1m)[r * 512 + tid];
  if (tid < 64) {
    reinterpret_cast<float *>(smem + B1_OFF)[tid] = bias1[tid] / w1inv;  // (exact: a power of two)
    reinterpret_cast<float *>(smem + B2_OFF)[tid] = bias2[tid];
  }
  for (int i = tid; i < 2 * RAW_BYTES / 8; i += 512) reinterpret_cast<u32x2 *>(smem + RAW_OFF)[i] = (u32x2){0u, 0u};

  // ---- raw patch: thread t < 288 owns pixel (t / 36, t % 36) of the 8 x 36 patch; the float planes arrive by LDS-DMA
  // into a staging area and are normalised a phase later (see conv1_regw_kernel) ----
  const int r_row = tid / 36, r_col = tid - r_row * 36;
  const bool r_mine = tid < 8 * 36;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  bool r_ok = false;
  auto raw_fetch = [&](int it) {
    r_ok = false;
    if (it < my_tiles && wave < STG_T / 64) {  // (wave-uniform: waves 0..4 hold the patch's 288 threads)
      int n, x0, y0;
      tile_coords(it, n, x0, y0);
      const int gy = y0 - 2 + r_row, gx = x0 - 2 + r_col;
      r_ok = r_mine && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const float *img = n < B ? x + (size_t)n * 3 * HW : y + (size_t)(n - B) * 3 * HW;
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, 3u * (unsigned)HW * 4u, 0x00020000);
      const unsigned off = r_ok ? (unsigned)((gy * W + gx) * 4) : 0x80000000u;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(smem + STG_OFF + c * (STG_T * 4) + wave * 256), 4,
                                                 off, c * HW * 4, 0, 0);
    }
  };
  static_assert(GPP * NPASS == 4, "raw_commit's counted wait assumes 4 stores per wave and tile");
  auto raw_commit = [&](int buf, bool stores_since) {
    if (stores_since)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (r_mine) {
      h4 hi, lo;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float raw = *reinterpret_cast<const float *>(smem + STG_OFF + (c * STG_T + tid) * 4);
        const float v = r_ok ? (raw - mean[c]) / sd[c] : 0.f;
        hi[c] = (_Float16)v;
        lo[c] = (_Float16)(v - (float)hi[c]);
      }
      hi[3] = lo[3] = (_Float16)0.f;
      char *p = smem + RAW_OFF + buf * RAW_BYTES + (r_row * RAWP + r_col) * 8;
      *reinterpret_cast<h4 *>(p) = hi;
      *reinterpret_cast<h4 *>(p + RAW_PLANE) = lo;
    }
  };
  // ---- conv1_1 of tile `it` (raw patch it&1) into halo image it&1: all waves, groups wave and wave+8 together ----
  auto conv1_1_halo = [&](int it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    char *slot = smem + (it & 1) * SLOT;
    const char *rawp = smem + RAW_OFF + (it & 1) * RAW_BYTES;
    constexpr int NU = (NGRP + 7) / 8;  // 2
    f32x4 a1[NU][4];
    int qv[NU], hyv[NU], hxv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = (wave + 8 * u) * 16 + l15, qc = q < NQ ? q : NQ - 1;
      qv[u] = q;
      hyv[u] = qc / HWD;
      hxv[u] = qc - hyv[u] * HWD;
#pragma unroll
      for (int i = 0; i < 4; ++i)  // the accumulator starts at bias x scale
        a1[u][i] = *reinterpret_cast<const f32x4 *>(smem + B1_OFF + (16 * i + 4 * c4) * 4);
    }
    const bool interior = y0 >= 1 && y0 + TH + 1 <= H && x0 >= 1 && x0 + TW + 1 <= W;  // (block-uniform)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u32x4 bh[NU], bl[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const char *rp = rawp + ((hyv[u] + 2 * m + (c4 >> 1)) * RAWP + hxv[u] + (c4 & 1) * 2) * 8;
        const u32x2 h0 = *reinterpret_cast<const u32x2 *>(rp), h1 = *reinterpret_cast<const u32x2 *>(rp + 8);
        const u32x2 l0 = *reinterpret_cast<const u32x2 *>(rp + RAW_PLANE), l1 = *reinterpret_cast<const u32x2 *>(rp + RAW_PLANE + 8);
        bh[u] = (u32x4){h0[0], h0[1], h1[0], h1[1]};
        bl[u] = (u32x4){l0[0], l0[1], l1[0], l1[1]};
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)  // w_hi*a_hi, w_lo*a_hi, w_hi*a_lo: consecutive MFMAs on different accumulators
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 wfr = *reinterpret_cast<const u32x4 *>(
              smem + W1_OFF + ((((i * 2 + m) * 2 + (term == 1 ? 1 : 0)) * 64 + lane) * 16));
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if (wave + 8 * u < NGRP)  // (wave-uniform)
              a1[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wfr),
                                                                __builtin_bit_cast(f16x8, term == 2 ? bl[u] : bh[u]),
                                                                a1[u][i], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = qv[u];
      const int gy = y0 - 1 + hyv[u], gx = x0 - 1 + hxv[u];
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      if (wave + 8 * u < NGRP && q < NQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // channels 16*i + 4*c4 .. +3: chunk i>>1, quarter 2*(i&1) + (c4>>1), half (c4&1)
          h4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (interior || inside) ? fmaxf(a1[u][i][e] * w1inv, 0.f) : 0.f;
            hi[e] = (_Float16)v;
            lo[e] = (_Float16)(v - (float)hi[e]);
          }
          char *p = slot + (i >> 1) * CH_BYTES + q * PITCH + ((2 * (i & 1) + (c4 >> 1)) << 4) + (c4 & 1) * 8;
          *reinterpret_cast<h4 *>(p) = hi;
          *reinterpret_cast<h4 *>(p + LO) = lo;
        }
      }
    }
  };

  const unsigned kOOB = 0x80000000u;
  const unsigned img_out_bytes = (unsigned)H * (unsigned)W * (unsigned)COUT * 4u;
  __syncthreads();  // zeroed patches, conv1_1 fragments and bias are in LDS
  raw_fetch(0);
  raw_commit(0, false);
  raw_fetch(1);
  raw_commit(1, false);
  __syncthreads();
  conv1_1_halo(0);
  auto conv1_2_tile = [&](int it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    const char *slot = smem + (it & 1) * SLOT;
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)n * H * W * COUT, 0, img_out_bytes, 0x00020000);
    // BOTH rows of the wave in one k loop (round 4): a step = one halo row hr (0..3 below the wave's first output row) x
    // one column tap kx x one 32-channel chunk; its four fragments (hi, lo x two 16-column groups) feed output row
    // r = hr - ky for every kernel row ky that exists -- 6 MFMAs in the first and last halo row, 12 in the middle two.
    // 24 steps and 96 fragment reads per tile instead of 36 and 144: with four waves of a CU in conv1_2 at once the
    // row-at-a-time form asked the LDS for 128 clocks of reads per 96 clocks of MFMA.
    u32x4 bh[2][GPP], bl[2][GPP];
    auto load_s = [&](const int (&q0)[GPP], int s, u32x4(&h)[GPP], u32x4(&l)[GPP]) {
      const int cc = s / 12, hr = (s - cc * 12) / 3, kx = s - cc * 12 - hr * 3;
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        h[g] = *reinterpret_cast<const u32x4 *>(slot + q0[g] + (cc * CH_BYTES + (hr * HWD + kx) * PITCH));
        l[g] = *reinterpret_cast<const u32x4 *>(slot + q0[g] + (cc * CH_BYTES + (hr * HWD + kx) * PITCH + LO));
      }
    };
    int q0[GPP];  // byte offset of this lane's hi fragment at halo row ph * RW, tap kx = 0, chunk 0, per 16-pixel group
#pragma unroll
    for (int g = 0; g < GPP; ++g) q0[g] = ((ph * RW) * HWD + g * 16 + l15) * PITCH + (c4 << 4);
    asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
    // hi*hi | the two cross terms w_lo*a_hi + w_hi*a_lo (2^-11 of the first: one accumulator, they are added in the end
    // anyway), per group and output row -- 32 registers; a third set spilled
    f32x4 acc[2][GPP][RW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[i][g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NSTEP = 2 * (RW + 2) * 3;
    load_s(q0, 0, bh[0], bl[0]);
#pragma unroll
    for (int s2 = 0; s2 < NSTEP; ++s2) {
      if (s2 + 1 < NSTEP) load_s(q0, s2 + 1, bh[(s2 + 1) & 1], bl[(s2 + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int cc = s2 / 12, hr = (s2 - cc * 12) / 3, kx = s2 - cc * 12 - hr * 3;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int ky = hr - r;
        if (ky < 0 || ky > 2) continue;
        const int ks = cc * 9 + ky * 3 + kx;
#pragma unroll
        for (int term = 0; term < 3; ++term)  // (a cross accumulator's two MFMAs are a term apart: never back to back)
#pragma unroll
          for (int g = 0; g < GPP; ++g)
            acc[term ? 1 : 0][g][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(f16x8, wf[term == 1 ? 1 : 0][ks]),
                __builtin_bit_cast(f16x8, term == 2 ? bl[s2 & 1][g] : bh[s2 & 1][g]), acc[term ? 1 : 0][g][r], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const int gy = y0 + ph * RW + r, gx = x0 + g * 16 + l15;
        const bool inside = gy < H && gx < W;
        const f32x4 bia = *reinterpret_cast<const f32x4 *>(smem + B2_OFF + (cg * 16 + 4 * c4) * 4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = fmaxf((acc[0][g][r][e] + acc[1][g][r][e]) * winv2 + bia[e], 0.f);
        const unsigned off = inside ? (unsigned)(((gy * W + gx) * COUT + cg * 16 + 4 * c4) * 4) : kOOB;
#ifdef NQA_SP_NO_STORE  // (timing-only: every store lands out of range)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off | kOOB, 0, 0);
#else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
#endif
      }
  };
  for (int it = 0; it < my_tiles; ++it) {
    __syncthreads();
    raw_fetch(it + 2);  // lands under this tile's MFMAs
    const bool next = it + 1 < my_tiles;
#if defined(NQA_SP_NO_P1)  // timing-only ablations (tools/gpu_split_ablate.sh; results are wrong on purpose)
    conv1_2_tile(it);
    (void)next;
#elif defined(NQA_SP_NO_P2)
    if (next) conv1_1_halo(it + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no stores in this ablation: the counted wait below would not hold)
#else
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {  // (a loop, so that each phase's code exists once)
      if ((half == 0) == (wave < 4)) {
        __builtin_amdgcn_s_setprio(0);
        conv1_2_tile(it);
      } else if (next) {
        __builtin_amdgcn_s_setprio(2);
        conv1_1_halo(it + 1);
      }
    }
#endif
    raw_commit(it & 1, true);  // tile it+2's patch; 4 stores younger than its pixels in either phase order (see conv1_regw_kernel)
  }
#endif
}

// ---------------------------------------------------------------------------------
// conv2_1 in f32s with the WEIGHTS IN REGISTERS (round 4): conv1_regw_split_kernel's conv1_2 on a halo that comes from
// memory.
// ---------------------------------------------------------------------------------
// The implicit GEMM runs this layer (64 -> 128 channels, 18 k-steps) at 0.42 of the issued-MFMA peak: its stages are
// shorter than an LDS-DMA round trip, as in the 16-bit modes before conv3x3_regw_kernel.  Here a wave keeps the (hi, lo)
// fragments of 16 output channels (144 VGPRs, the two-term blob format of conv1_2), a block of 8 waves covers 64 output
// channels x a 4 x 32 tile (4 channel groups x 2 row pairs), and the two 64-channel halves of the layer are separate
// blocks that share an XCD and walk the same tile sequence (the second one finds the halo in L2).  The input is the
// split16 pooled tap (256 B per pixel); LDS-DMA brings the 6 x 34 halo patch in as the records conv1_2 reads --
// [32 channels hi | 32 channels lo | 32 B pad] per 32-channel chunk, 160 B, conflict-free for a ds_read_b128 of 16
// consecutive pixels -- by giving each 16-byte item its own global address (the split16 record interleaves hi and lo
// per 8 channels; the permutation costs nothing).  Two slots (130 560 B), the next tile's patch requested one tile
// ahead, one barrier and one counted vmcnt per tile.  The k loop is conv1_2's (both rows of the wave per step, the
// cross terms in one accumulator); split16 records out (conv2_1 is not a tapped layer).
__global__ __launch_bounds__(512) void conv3x3_regw_split_kernel(const char *__restrict__ in, const char *__restrict__ wreg,
                                                                 const float *__restrict__ bias, char *__restrict__ out,
                                                                 int H, int W, int tiles_x, int tiles_y, int total_tiles,
                                                                 int cout) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CIN = 64, TH = 4, TW = 32, HWD = TW + 2, NQ = (TH + 2) * HWD;  // 204 halo pixels
  constexpr int PITCH = 160, LO = 64, CH_BYTES = NQ * PITCH, SLOT = 2 * CH_BYTES;
  constexpr int ITEMS = 2 * NQ * (PITCH / 16), ROUNDS = (ITEMS + 511) / 512;  // 4 080 sixteen-byte items, 8 DMA rounds
  constexpr int RW = 2, GPP = 2, NSTORE = 2 * RW * GPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [halo slot 0][halo slot 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, c4 = lane >> 4;
  const int cg = wave & 3, ph = wave >> 2;
  const int nh = cout >> 6;  // 64-channel halves of the layer (conv2_1: 2)

  // blocks with the same (XCD, stream) and different halves walk the same tiles
  const int nblk = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int half = jb % nh, stream = jb / nh, streams = (nblk >> 3) / nh;
  const int t_lo = (int)((long)total_tiles * xcd / 8), t_hi = (int)((long)total_tiles * (xcd + 1) / 8);
  const int my_tiles = t_lo + stream < t_hi ? (t_hi - t_lo - stream - 1) / streams + 1 : 0;
  if (my_tiles == 0) return;  // (block-uniform)
  auto tile_coords = [&](int it, int &n, int &x0, int &y0) {
    const int t = t_lo + stream + it * streams;
    n = t / (tiles_x * tiles_y);
    const int t2 = t - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * TW;
    y0 = by * TH;
  };

  u32x4 wf[2][18];  // [hi | lo][k-step] of this wave's 16 output channels
  const int cgg = half * 4 + cg;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      wf[i][ks] = *reinterpret_cast<const u32x4 *>(wreg + ((((size_t)cgg * 2 + i) * 18 + ks) * 64 + lane) * 16);
  const f32x4 bia = *reinterpret_cast<const f32x4 *>(bias + cgg * 16 + 4 * c4);
  const float winv = bias[cout];

  // DMA plan: item j = round * 512 + tid is 16 bytes of LDS; record j / 10 = (chunk, halo pixel), part j % 10: parts 0..3
  // the chunk's hi halves of channels 8 p .., parts 4..7 its lo halves, 8..9 the pad.  Packed per round: halo row (4 bits),
  // halo column (6 bits), byte offset inside the pixel's split16 record (9 bits), bit 19 = a live item.
  int plan[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int j = r * 512 + tid, rec = j / 10, part = j - rec * 10;
    const int cc = rec >= NQ ? 1 : 0, q = rec - cc * NQ, hy = q / HWD, hx = q - hy * HWD;
    const int p4 = part & 3, piece = (p4 >> 1) * 4 + (p4 & 1) + (part >= 4 ? 2 : 0);  // 16-byte piece of the 128-byte chunk
    const bool live = j < ITEMS && part < 8;
    plan[r] = live ? ((hy & 15) | (hx << 4) | ((cc * 128 + piece * 16) << 10) | (1 << 19)) : 0;
  }
  const unsigned kOOB = 0x80000000u;
  const unsigned img_in_bytes = (unsigned)H * (unsigned)W * (unsigned)(CIN * 4);
  const unsigned img_out_bytes = (unsigned)H * (unsigned)W * (unsigned)cout * 4u;
  auto issue_halo = [&](int it, int slot_idx) {
    char *slot = smem + slot_idx * SLOT;
    const bool real = it < my_tiles;  // past the last tile: the same number of pieces, all out of range
    int n = 0, x0 = 0, y0 = 0;
    if (real) tile_coords(it, n, x0, y0);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(in + (size_t)n * H * W * (CIN * 4)), 0, img_in_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int pl = plan[r];
      const int gy = y0 - 1 + (pl & 15), gx = x0 - 1 + ((pl >> 4) & 63);
      const bool live = real && (pl >> 19);
      const bool ok = live && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)((gy * W + gx) * (CIN * 4) + ((pl >> 10) & 511)) : kOOB;
      // (the last 16 lanes of the last round would address LDS BEHIND the slot -- record (0, 0) of the other slot, which
      // the block is reading at that moment.  An out-of-range offset is not enough there: with those lanes left in the
      // instruction single pixels came out wrong in ~1 of 2 launches on ragged maps, tools/gpu_regw_split_check.py;
      // masked out of EXEC the instruction still counts once in vmcnt)
      if (r * 512 + tid < ITEMS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(slot + r * 8192 + wave * 1024), 16, off, 0, 0, 0);
      if (live && !ok)  // zero padding (the DMA transfers nothing for an out-of-range lane)
        *reinterpret_cast<u32x4 *>(slot + (r * 512 + tid) * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  };

  issue_halo(0, 0);
  for (int it = 0; it < my_tiles; ++it) {
    int n, x0, y0;
    tile_coords(it, n, x0, y0);
    // tile `it`'s patch was requested before the previous tile's stores: everything but those has retired
    if (it == 0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
    issue_halo(it + 1, (it + 1) & 1);  // into the slot tile it - 1 read, which every wave left before the barrier
    const char *slot = smem + (it & 1) * SLOT;
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)n * H * W * cout * 4, 0, img_out_bytes, 0x00020000);
    u32x4 bh[2][GPP], bl[2][GPP];
    auto load_s = [&](const int (&q0)[GPP], int s, u32x4(&h)[GPP], u32x4(&l)[GPP]) {
      const int cc = s / 12, hr = (s - cc * 12) / 3, kx = s - cc * 12 - hr * 3;
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        h[g] = *reinterpret_cast<const u32x4 *>(slot + q0[g] + (cc * CH_BYTES + (hr * HWD + kx) * PITCH));
        l[g] = *reinterpret_cast<const u32x4 *>(slot + q0[g] + (cc * CH_BYTES + (hr * HWD + kx) * PITCH + LO));
      }
    };
    int q0[GPP];
#pragma unroll
    for (int g = 0; g < GPP; ++g) q0[g] = ((ph * RW) * HWD + g * 16 + l15) * PITCH + (c4 << 4);
    asm volatile("" : "+v"(q0[0]), "+v"(q0[1]));
    f32x4 acc[2][GPP][RW];  // hi*hi | the two cross terms, per group and output row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < GPP; ++g)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[i][g][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NSTEP = 2 * (RW + 2) * 3;
    load_s(q0, 0, bh[0], bl[0]);
#pragma unroll
    for (int s2 = 0; s2 < NSTEP; ++s2) {
      if (s2 + 1 < NSTEP) load_s(q0, s2 + 1, bh[(s2 + 1) & 1], bl[(s2 + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int cc = s2 / 12, hr = (s2 - cc * 12) / 3, kx = s2 - cc * 12 - hr * 3;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int ky = hr - r;
        if (ky < 0 || ky > 2) continue;
        const int ks = cc * 9 + ky * 3 + kx;
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int g = 0; g < GPP; ++g)
            acc[term ? 1 : 0][g][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(f16x8, wf[term == 1 ? 1 : 0][ks]),
                __builtin_bit_cast(f16x8, term == 2 ? bl[s2 & 1][g] : bh[s2 & 1][g]), acc[term ? 1 : 0][g][r], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // descale, bias, ReLU; channels c = 16 cgg + 4 c4 .. + 3 of a pixel's split16 record: hi halves at
    // (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2, lo halves 32 bytes behind (store_split4)
    const unsigned rec_off = (unsigned)(cgg * 64 + (c4 >> 1) * 16 + (c4 & 1) * 8);
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int g = 0; g < GPP; ++g) {
        const int gy = y0 + ph * RW + r, gx = x0 + g * 16 + l15;
        const bool inside = gy < H && gx < W;
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaxf((acc[0][g][r][e] + acc[1][g][r][e]) * winv + bia[e], 0.f);
          hi[e] = (_Float16)v;
          lo[e] = (_Float16)(v - (float)hi[e]);
        }
        const unsigned off = inside ? (unsigned)(gy * W + gx) * (unsigned)(cout * 4) + rec_off : kOOB;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), orsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), orsrc, off, 32, 0);
      }
  }
#endif
}

// ---------------------------------------------------------------------------------
// conv1_1 + conv1_2 fused (16-bit modes): the whole of stage 1 without the 64-channel
// full-resolution intermediate ever touching HBM.
// ---------------------------------------------------------------------------------
// Persistent 8-wave blocks (one per CU) keep ALL of conv1_2's weights in LDS (9 taps x 64 x 64 x
// 2 B = 72 KB) and walk over 4x32-pixel output tiles.  A tile goes through two phases:
//   P1  conv1_1 for its 6x34 halo patch.  Per 32-pixel column tile a wave fetches the 4x36x3 raw
//       pixels it needs (one work item ahead, into registers), writes them normalised
//       ((x-mean)/std, zero outside the image = conv1_1's padding) as 16-bit [row][col][4] into
//       its private LDS scratch and runs conv1_1 on MFMA: with the contraction index ordered
//       k = ky*16 + kx*4 + c (kx, c padded to 4 with zero weights) a lane's 8-element B fragment
//       is 16 contiguous bytes of that scratch, so K = 48 is three MFMAs per 32x32 tile.
//       bias+ReLU, zero outside the image (conv1_2's padding), stored straight into the swizzled
//       [chunk][pixel][64 B] image that P2 reads;
//   P2  conv1_2: the 6-stage loop of conv3x3_igemm_kernel with nothing left to stream (no DMA, no
//       barrier inside), bias+ReLU, NHWC store.
// The block's two wave groups (waves 0-3, 4-7) each own a tile stream and run half a period
// apart: while one group is in P2 (MFMA-bound) the other is in P1 (VALU/LDS-bound), so the two
// waves of every SIMD complement each other.  One block barrier per half period.
struct Conv1Fused {
  typedef ConvGeom<1, 4, 2, 1, 32> G;  // per group: 64 ch x 128 px over 4 waves
  static constexpr int A_TILE = 2 * G::A_BYTES;  // both 32-channel chunks of one tile
  static constexpr int W2_BYTES = 2 * 9 * 64 * 64;
  static constexpr int RAW_W = 40, RAW_WAVE_BYTES = 4 * RAW_W * 8;
  static constexpr int W1_BYTES = 3 * 64 * 2 * 16;  // conv1_1 as MFMA A fragments [ky][cout][h][8]
  static constexpr int A_OFF = 0, W2_OFF = 2 * A_TILE, RAW_OFF = W2_OFF + W2_BYTES;
  static constexpr int W1_OFF = RAW_OFF + 8 * RAW_WAVE_BYTES, BIAS_OFF = W1_OFF + W1_BYTES;
  static constexpr int LDS_BYTES = BIAS_OFF + 2 * 64 * 4;
  static constexpr int NPT = (G::NQ + 31) / 32;  // 32-pixel column tiles of the halo patch (7)
};

template <typename P>
__global__ __launch_bounds__(512) void conv1_fused_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          int B, const char *__restrict__ w1m,
                                                          const float *__restrict__ bias1,
                                                          const char *__restrict__ w2pk,
                                                          const float *__restrict__ bias2,
                                                          typename P::T *__restrict__ out, int H, int W, int tiles_x,
                                                          int tiles_y, int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename P::T T;
  typedef Conv1Fused F;
  typedef F::G G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int HW = H * W;
  const int grp = wave >> 2, w4 = wave & 3;
  char *abuf = smem + F::A_OFF + grp * F::A_TILE;
  char *raw = smem + F::RAW_OFF + wave * F::RAW_WAVE_BYTES;

  // ---- one-time: zero the raw scratch (4th channel / pad columns stay 0), fetch the weights ----
  for (int i = tid; i < 8 * F::RAW_WAVE_BYTES / 16; i += 512) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4 *>(smem + F::RAW_OFF)[i] = z;
  }
  {
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(w2pk), 0, F::W2_BYTES, 0x00020000);
#pragma unroll
    for (int r = 0; r < F::W2_BYTES / 16 / 512; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t *)(smem + F::W2_OFF + (r * 512 + wave * 64) * 16),
                                               16, (r * 512 + tid) * 16, 0, 0, 0);
  }
  if (tid < F::W1_BYTES / 16)
    reinterpret_cast<u32x4 *>(smem + F::W1_OFF)[tid] = reinterpret_cast<const u32x4 *>(w1m)[tid];
  if (tid < 128) reinterpret_cast<float *>(smem + F::BIAS_OFF)[tid] = tid < 64 ? bias1[tid] : bias2[tid - 64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto tile_coords = [&](int tile, int &n, int &x0, int &y0) {
    n = tile / (tiles_x * tiles_y);
    const int t2 = tile - n * (tiles_x * tiles_y), by = t2 / tiles_x;
    x0 = (t2 - by * tiles_x) * 32;
    y0 = by * 4;
  };

  // ---- P1 pieces: a wave owns column tiles pt = w4 and w4+4 (< 7) of its group's halo patch ----
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  int r_plan[7];  // item -> (row << 16 | col << 2 | c) of a 4x36x3 raw sub-patch, -1 = none
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int i = r * 64 + lane;
    const int row = i / 108, rem = i - row * 108, col = rem / 3;
    r_plan[r] = i < 432 ? (row << 16 | col << 2 | (rem - col * 3)) : -1;
  }
  // per-lane part of a raw pixel's address, fixed for the whole kernel: c*HW + row*W + col
  int r_off[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int row = r_plan[r] >> 16, col = (r_plan[r] >> 2) & 0x3FFF, c = r_plan[r] & 3;
    r_off[r] = c * HW + row * W + col;
  }
  auto fetch = [&](int n, int x0, int y0, int pt, float(&rv)[7], unsigned &okmask) {
    const float *img = (n < B ? x + (size_t)n * 3 * HW : y + (size_t)(n - B) * 3 * HW);
    const int r0 = (pt * 32) / G::HW_;  // first halo row this column tile touches
    const int gy0 = y0 - 2 + r0, gx0 = x0 - 2;
    const float *org = img + (gy0 * W + gx0);  // may point before the image: only dereferenced when ok
    okmask = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int row = r_plan[r] >> 16, col = (r_plan[r] >> 2) & 0x3FFF;
      const bool ok = r_plan[r] >= 0 && (unsigned)(gy0 + row) < (unsigned)H && (unsigned)(gx0 + col) < (unsigned)W;
      rv[r] = 0.f;
#ifndef NQA_F_NO_FETCH
      if (ok) rv[r] = org[r_off[r]];
#endif
      okmask |= ok ? (1u << r) : 0u;
    }
  };
  auto finish = [&](int x0, int y0, int pt, const float(&rv)[7], unsigned okmask) {
#ifdef NQA_F_NO_P1
    asm volatile("" ::"v"(rv[0]), "v"(rv[1]), "v"(rv[2]), "v"(rv[3]), "v"(rv[4]), "v"(rv[5]), "v"(rv[6]), "v"(okmask));
    return;
#endif
    const int r0 = (pt * 32) / G::HW_;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int row = r_plan[r] >> 16, col = (r_plan[r] >> 2) & 0x3FFF, c = r_plan[r] & 3;
      const float mu = c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2], sg = c == 0 ? sd[0] : c == 1 ? sd[1] : sd[2];
      const float v = (okmask >> r) & 1u ? (rv[r] - mu) / sg : 0.f;
      if (r_plan[r] >= 0) *reinterpret_cast<T *>(raw + (row * F::RAW_W + col) * 8 + c * 2) = P::from_f(v);
    }
    // the scratch is private to this wave: its own LDS writes are ordered before its reads
    const int q = pt * 32 + l31;
    const int qc = q < G::NQ ? q : G::NQ - 1;
    const int hy = qc / G::HW_, hx = qc - hy * G::HW_;
    f32x16 a1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[i][r] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const char *rp = raw + ((hy - r0 + ky) * F::RAW_W + hx + 2 * h) * 8;
      const u32x2 lo = *reinterpret_cast<const u32x2 *>(rp), hi = *reinterpret_cast<const u32x2 *>(rp + 8);
      const u32x4 bfr = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 wf = *reinterpret_cast<const u32x4 *>(smem + F::W1_OFF + ((ky * 64 + i * 32 + l31) * 2 + h) * 16);
        a1[i] = P::mma(wf, bfr, a1[i]);
      }
    }
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    if (q < G::NQ) {
      const int sw = (q >> 2) & 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(smem + F::BIAS_OFF + (i * 32 + 8 * g + 4 * h) * 4);
          typedef __attribute__((ext_vector_type(4))) T t4;
          t4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = P::from_f(inside ? fmaxf(a1[i][4 * g + e] + b4[e], 0.f) : 0.f);
          *reinterpret_cast<t4 *>(abuf + i * G::A_BYTES + q * 64 + ((g ^ sw) << 4) + h * 8) = v;
        }
      }
    }
  };
  float rawA[7], rawB[7];
  unsigned okA = 0, okB = 0;
  const bool hasB = w4 + 4 < F::NPT;
  // raw pixels of both of this wave's column tiles of `tile` -> registers (no wait)
  auto fetch_both = [&](int tile) {
    int n, x0, y0;
    tile_coords(tile, n, x0, y0);
    fetch(n, x0, y0, w4, rawA, okA);
    if (hasB) fetch(n, x0, y0, w4 + 4, rawB, okB);
  };
  // precondition: fetch_both(tile) was issued (a phase earlier, so the loads have landed)
  auto phase1 = [&](int tile) {
    int n, x0, y0;
    tile_coords(tile, n, x0, y0);
    finish(x0, y0, w4, rawA, okA);
    if (hasB) finish(x0, y0, w4 + 4, rawB, okB);
  };

  // ---- P2: conv1_2 of the group's tile; wave w4 owns tile row w4 ----
  int w_base[2], w_sw[2];
  const int q0 = w4 * G::HW_ + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r64 = i * 32 + l31;
    w_base[i] = F::W2_OFF + r64 * 64;
    w_sw[i] = (r64 >> 2) & 3;
  }
  // `prefetch_tile` (< 0: none): the tile whose P1 this wave runs next; its first raw fetch is
  // issued here so that it lands under the MFMAs
  auto phase2 = [&](int tile, int prefetch_tile) {
    int n, x0, y0;
    tile_coords(tile, n, x0, y0);
    if (prefetch_tile >= 0) fetch_both(prefetch_tile);
    // make the per-lane bases opaque per tile: otherwise hipcc hoists all 108 fragment addresses
    // of the unrolled loop out of the persistent tile loop and runs out of registers
    int wb[2] = {w_base[0], w_base[1]}, qq = q0;
    asm volatile("" : "+v"(wb[0]), "+v"(wb[1]), "+v"(qq));
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto load_frags = [&](int t, u32x4(&af)[2], u32x4 &bf) {  // t = 0..35: (cc, ky, kx, ks)
      const int st = t / 6, tt = t - st * 6, cc = st / 3, ky = st - cc * 3;
      const int kx = tt >> 1, ch = 2 * (tt & 1) + h;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const u32x4 *>(smem + wb[i] + st * 12288 + kx * 4096 + ((ch ^ w_sw[i]) << 4));
      const int q = qq + ky * G::HW_ + kx;
      bf = *reinterpret_cast<const u32x4 *>(abuf + cc * G::A_BYTES + q * 64 + ((ch ^ ((q >> 2) & 3)) << 4));
    };
    // only two MFMAs per k-step here, so fragments are read TWO steps ahead (three register sets)
    u32x4 af0[2], af1[2], af2[2], bf0, bf1, bf2;
    load_frags(0, af0, bf0);
#ifdef NQA_F_NO_P2LOOP
    acc[0] = P::mma(af0[0], bf0, acc[0]);
    acc[1] = P::mma(af0[1], bf0, acc[1]);
#else
    load_frags(1, af1, bf1);
#pragma unroll
    for (int t = 0; t < 36; t += 3) {
      load_frags(t + 2, af2, bf2);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = P::mma(af0[0], bf0, acc[0]);
      acc[1] = P::mma(af0[1], bf0, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 3 < 36) load_frags(t + 3, af0, bf0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = P::mma(af1[0], bf1, acc[0]);
      acc[1] = P::mma(af1[1], bf1, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 4 < 36) load_frags(t + 4, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = P::mma(af2[0], bf2, acc[0]);
      acc[1] = P::mma(af2[1], bf2, acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    const int gy = y0 + w4, gx = x0 + l31;
#ifdef NQA_F_NO_STORE
    if (gy < -5) {
#else
    if (gy < H && gx < W) {
#endif
      T *o = out + ((size_t)(n * H + gy) * W + gx) * 64 + 4 * h;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(smem + F::BIAS_OFF + (64 + i * 32 + 8 * g + 4 * h) * 4);
          typedef __attribute__((ext_vector_type(4))) T t4;
          t4 s4;
#pragma unroll
          for (int e = 0; e < 4; ++e) s4[e] = P::from_f(fmaxf(acc[i][4 * g + e] + b4[e], 0.f));
          *reinterpret_cast<t4 *>(o + i * 32 + 8 * g) = s4;
        }
      }
    }
  };

  // ---- schedule: group g's k-th tile is (2*blockIdx.x + g) + k * 2*gridDim.x ----
  const int stride = 2 * gridDim.x;
  const int first = 2 * blockIdx.x + grp;
  const int K = first < total_tiles ? (total_tiles - 1 - first) / stride + 1 : 0;       // my tiles
  const int K0 = 2 * (int)blockIdx.x < total_tiles ? (total_tiles - 1 - 2 * (int)blockIdx.x) / stride + 1 : 0;  // group 0's
  if (K > 0) fetch_both(first);
  __syncthreads();  // weights, biases and the zeroed scratch are visible
  if (grp == 0 && K > 0) phase1(first);
  __syncthreads();
  for (int k = 0; k < K0; ++k) {  // K0 >= group 1's count, so both groups see the same barriers
    const int t_k = first + k * stride, t_n = t_k + stride;
    if (grp == 0) {
      if (k < K) phase2(t_k, k + 1 < K ? t_n : -1);
    } else {
      if (k < K) phase1(t_k);
    }
    __syncthreads();
    if (grp == 0) {
      if (k + 1 < K) phase1(t_n);
    } else {
      if (k < K) phase2(t_k, k + 1 < K ? t_n : -1);
    }
    __syncthreads();
  }
#endif
}

// ---------------------------------------------------------------------------------
// Stage 1, second form (16-bit modes): conv1_2 as an ordinary implicit-GEMM tile whose halo is not
// loaded but COMPUTED -- the block runs conv1_1 (+ bias, ReLU, conv1_2's zero padding) for its
// 10x34 halo patch straight into the two LDS halo buffers (64 channels = both 32-channel chunks),
// then streams only conv1_2's weights through the usual 6-stage loop.  relu1_1 never exists in HBM
// and nothing of the tile waits for an activation DMA; two independent blocks per CU overlap one
// block's conv1_1 prologue (loads, VALU, LDS writes) with the other's MFMA stages.
// conv1_1 runs on MFMA as in conv1_fused_kernel: k = ky*16 + kx*4 + c, so a lane's B fragment is
// 16 contiguous bytes of the wave's normalised raw patch and K = 48 is three MFMAs per 32x32 tile.
struct Conv1Tile {
  typedef ConvGeom<1, 4, 2, 2, 32> G;  // 64 ch x 256 px (8 x 32), 4 waves, each 64 ch x 64 px
  static constexpr int RAW_W = 40, RAW_WAVE_BYTES = 4 * RAW_W * 8;
  static constexpr int RAW_OFF = G::LDS_BYTES;
  static constexpr int LDS_BYTES = RAW_OFF + 4 * RAW_WAVE_BYTES;
  static constexpr int NPT = (G::NQ + 31) / 32;  // 32-pixel column tiles of the halo patch (11)
  static constexpr int PT_PER_WAVE = (NPT + 3) / 4;
};

template <typename P>
__global__ __launch_bounds__(256) void conv1_tile_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         int B, const char *__restrict__ w1m,
                                                         const float *__restrict__ bias1,
                                                         const char *__restrict__ w2pk,
                                                         const float *__restrict__ bias2,
                                                         typename P::T *__restrict__ out, int H, int W, int tiles_x) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename P::T T;
  typedef Conv1Tile F;
  typedef F::G G;
  constexpr int WN_T = 2, WM_T = 2, TW = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A0][A1][W0][W1][raw x4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int HW = H * W;
  int tile_id = blockIdx.x;
  {
    const int nb = gridDim.x, qq = nb >> 3, rr = nb & 7, xcd = tile_id & 7, local = tile_id >> 3;
    tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
  }
  const int bx = tile_id % tiles_x, by = tile_id / tiles_x;
  const int n = blockIdx.y;
  const int x0 = bx * TW, y0 = by * G::TH;
  const int wm = wave;
  constexpr int S = 6;  // 2 chunks of 32 channels x 3 kernel rows

  // ---- weights of conv1_2: LDS-DMA, one kernel row of one chunk per stage ----
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(w2pk), 0, 2 * 9 * 64 * 64, 0x00020000);
  unsigned w_goff[G::W_ROUNDS];
#pragma unroll
  for (int r = 0; r < G::W_ROUNDS; ++r) w_goff[r] = (unsigned)(r * G::THREADS + tid) * 16u;
  const int wave_base = wave * 64 * 16;
  auto issue_w = [&](int s) {
    char *wdst = smem + 2 * G::A_BYTES + (s & 1) * G::W_BYTES + wave_base;
#pragma unroll
    for (int r = 0; r < G::W_ROUNDS; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t *)(wdst + r * G::THREADS * 16), 16, w_goff[r],
                                               s * (G::SUB_STAGE_ITEMS * 16), 0, 0);
  };
  issue_w(0);  // lands under the conv1_1 prologue

  // ---- prologue: conv1_1 of the halo patch; a wave owns column tiles pt = wave, wave+4, wave+8 ----
  char *raw = smem + F::RAW_OFF + wave * F::RAW_WAVE_BYTES;
  for (int i = lane; i < F::RAW_WAVE_BYTES / 16; i += 64) {  // 4th channel / pad columns stay 0
    const u32x4 z = {0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4 *>(raw)[i] = z;
  }
  u32x4 w1f[3][2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      w1f[ky][i] = *reinterpret_cast<const u32x4 *>(w1m + ((ky * 64 + i * 32 + l31) * 2 + h) * 16);
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  int r_plan[7], r_off[7];  // item -> (row << 16 | col << 2 | c) of a 4x36x3 raw sub-patch, -1 = none
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int i = r * 64 + lane;
    const int row = i / 108, rem = i - row * 108, col = rem / 3, c = rem - col * 3;
    r_plan[r] = i < 432 ? (row << 16 | col << 2 | c) : -1;
    r_off[r] = c * HW + row * W + col;
  }
  const float *img = (n < B ? x + (size_t)n * 3 * HW : y + (size_t)(n - B) * 3 * HW);
  float rv[F::PT_PER_WAVE][7];
  unsigned okm[F::PT_PER_WAVE];
#pragma unroll
  for (int u = 0; u < F::PT_PER_WAVE; ++u) {
    const int pt = wave + 4 * u;
    okm[u] = 0;
    if (pt < F::NPT) {  // wave-uniform
      const int r0 = (pt * 32) / G::HW_;
      const int gy0 = y0 - 2 + r0, gx0 = x0 - 2;
      const float *org = img + (gy0 * W + gx0);  // may point before the image: only dereferenced when ok
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        const int row = r_plan[r] >> 16, col = (r_plan[r] >> 2) & 0x3FFF;
        const bool ok = r_plan[r] >= 0 && (unsigned)(gy0 + row) < (unsigned)H && (unsigned)(gx0 + col) < (unsigned)W;
#ifdef NQA_T_NO_FETCH
        rv[u][r] = 0.f;
#else
        rv[u][r] = ok ? org[r_off[r]] : 0.f;
#endif
        okm[u] |= ok ? (1u << r) : 0u;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < F::PT_PER_WAVE; ++u) {
    const int pt = wave + 4 * u;
#ifdef NQA_T_NO_P1
    asm volatile("" ::"v"(rv[u][0]), "v"(rv[u][3]), "v"(rv[u][6]), "v"(okm[u]));
    if (false) {
#else
    if (pt < F::NPT) {
#endif
      const int r0 = (pt * 32) / G::HW_;
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        const int row = r_plan[r] >> 16, col = (r_plan[r] >> 2) & 0x3FFF, c = r_plan[r] & 3;
        const float mu = c == 0 ? mean[0] : c == 1 ? mean[1] : mean[2], sg = c == 0 ? sd[0] : c == 1 ? sd[1] : sd[2];
        const float v = (okm[u] >> r) & 1u ? (rv[u][r] - mu) / sg : 0.f;
        if (r_plan[r] >= 0) *reinterpret_cast<T *>(raw + (row * F::RAW_W + col) * 8 + c * 2) = P::from_f(v);
      }
      // the scratch is private to this wave: its own LDS writes are ordered before its reads
      const int q = pt * 32 + l31;
      const int qc = q < G::NQ ? q : G::NQ - 1;
      const int hy = qc / G::HW_, hx = qc - hy * G::HW_;
      f32x16 a1[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[i][r] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char *rp = raw + ((hy - r0 + ky) * F::RAW_W + hx + 2 * h) * 8;
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(rp), hi = *reinterpret_cast<const u32x2 *>(rp + 8);
        const u32x4 bfr = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
        for (int i = 0; i < 2; ++i) a1[i] = P::mma(w1f[ky][i], bfr, a1[i]);
      }
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      if (q < G::NQ) {
        const int sw = (q >> 2) & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias1 + i * 32 + 8 * g + 4 * h);
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = P::from_f(inside ? fmaxf(a1[i][4 * g + e] + b4[e], 0.f) : 0.f);
            *reinterpret_cast<t4 *>(smem + i * G::A_BYTES + q * 64 + ((g ^ sw) << 4) + h * 8) = v;
          }
        }
      }
    }
  }

  // ---- conv1_2: the stage loop of conv3x3_igemm_kernel (32x32x16 path) with nothing but weights to stream ----
  int w_base[WN_T], w_sw[WN_T];
#pragma unroll
  for (int i = 0; i < WN_T; ++i) {
    const int r64 = i * 32 + l31;
    w_base[i] = 2 * G::A_BYTES + r64 * 64;
    w_sw[i] = (r64 >> 2) & 3;
  }
  int q0[WM_T];
#pragma unroll
  for (int j = 0; j < WM_T; ++j) {
    const int m = (wm * WM_T + j) * 32 + l31;
    const int ty = m / TW, tx = m - ty * TW;
    q0[j] = ty * G::HW_ + tx;
  }
  f32x16 acc[WN_T][WM_T];
#pragma unroll
  for (int i = 0; i < WN_T; ++i)
#pragma unroll
    for (int j = 0; j < WM_T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef NQA_T_NO_LOOP
  for (int s = 0; s < 1; ++s) {
#else
  for (int s = 0; s < S; ++s) {
#endif
    const int cc = s / 3, ky = s - cc * 3;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // stage s's weights (and, at s = 0, every wave's part of the halo) are visible
    if (s + 1 < S) issue_w(s + 1);
    const char *abuf = smem + cc * G::A_BYTES;
    const char *wbuf = smem + (s & 1) * G::W_BYTES;
    auto load_frags = [&](int t, u32x4(&af)[WN_T], u32x4(&bf)[WM_T]) {
      const int kx = t >> 1, ch = 2 * (t & 1) + h;
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
        af[i] = *reinterpret_cast<const u32x4 *>(wbuf + w_base[i] + kx * 4096 + ((ch ^ w_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < WM_T; ++j) {
        const int q = q0[j] + ky * G::HW_ + kx;
        bf[j] = *reinterpret_cast<const u32x4 *>(abuf + q * 64 + ((ch ^ ((q >> 2) & 3)) << 4));
      }
    };
    auto mma_all = [&](const u32x4(&af)[WN_T], const u32x4(&bf)[WM_T]) {
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
#pragma unroll
        for (int j = 0; j < WM_T; ++j) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
    };
    u32x4 afA[WN_T], bfA[WM_T], afB[WN_T], bfB[WM_T];
    load_frags(0, afA, bfA);
#pragma unroll
    for (int t = 0; t < 6; t += 2) {
      load_frags(t + 1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < 6) load_frags(t + 2, afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: bias + ReLU, staged through LDS as in conv3x3_igemm_kernel ----
  constexpr int RB = 64 * (int)sizeof(T), NCH = RB / 16, SWZ = NCH - 1, ROWS = 4 * 32;
  const size_t rec = (size_t)64 * sizeof(T);
  char *const obase = reinterpret_cast<char *>(out);
#pragma unroll
  for (int j = 0; j < WM_T; ++j) {
    __syncthreads();
    {
      const int row = wm * 32 + l31;
      char *const rbase = smem + row * RB;
      const int sw = row & SWZ;
#pragma unroll
      for (int i = 0; i < WN_T; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cl = i * 32 + 8 * g + 4 * h;
          const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias2 + cl);
          typedef __attribute__((ext_vector_type(4))) T t4;
          t4 s4;
#pragma unroll
          for (int e = 0; e < 4; ++e) s4[e] = P::from_f(fmaxf(acc[i][j][4 * g + e] + b4[e], 0.f));
          *reinterpret_cast<t4 *>(rbase + (((cl >> 3) ^ sw) << 4) + (cl & 4) * 2) = s4;
        }
    }
    __syncthreads();
#pragma unroll 2
    for (int idx = tid; idx < ROWS * NCH; idx += G::THREADS) {
      const int row = idx / NCH, k = idx - row * NCH;
      const int m = ((row >> 5) * WM_T + j) * 32 + (row & 31);
      const int ty = m / TW, tx = m - ty * TW;
      const int gy = y0 + ty, gx = x0 + tx;
      if (gy < H && gx < W)
        *reinterpret_cast<u32x4 *>(obase + ((size_t)(n * H + gy) * W + gx) * rec + k * 16) =
            *reinterpret_cast<const u32x4 *>(smem + row * RB + ((k ^ (row & SWZ)) << 4));
    }
  }
#endif
}

// ---------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------
#ifdef NQA_STAMPS
extern "C" int nqa_debug_stamps(unsigned long long *out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

static int current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

// thread-local: a tuning choice made by one thread (tools timing variants against each other) never changes what
// another thread's calls launch
static thread_local int g_conv_variant = 1;  // 0: 4-wave tiles everywhere; 1: + 8-wave 256x256 tiles; 2: + 8-wave 128x512 tiles
static thread_local int g_stage1_variant = 0;  // 0: persistent two-phase kernel (conv1_fused_kernel); 1: conv1_tile_kernel
static thread_local int g_no_regw128 = 0;  // 1: conv2_2 / conv3_1 on the implicit GEMM instead of the register-weights kernel (A/B timing)
static thread_local int g_first_forms = 0;  // 1: the round-1 forms of stage 1 (two-phase kernel) and conv2_1 (implicit GEMM), for A/B timing
void set_conv_variant(int v) {
  g_conv_variant = v & 3;
  g_stage1_variant = (v >> 2) & 1;
}
bool mixed_stage1_unfused() { return g_first_forms != 0; }
void set_conv_first_forms(int on) {
  g_first_forms = on & 1;
  g_no_regw128 = (on >> 1) & 1;
}

template <typename P>
static int launch_conv1_1(const float *x, int n, int H, int W, const char *packed, void *out, hipStream_t st,
                          int blob_prec = P::ID) {
  const float *w = reinterpret_cast<const float *>(packed + layer_offset(0, blob_prec));
  const float *b = reinterpret_cast<const float *>(packed + layer_bias_offset(0, blob_prec));
  dim3 grid(cdiv(H * W, 256), n);
  TimedLaunch t(NQA_K_CONV1, st);
  conv1_1_kernel<P><<<grid, 256, 0, st>>>(x, w, b, reinterpret_cast<typename P::T *>(out), H, W);
  return check_launch("conv1_1");
}

template <typename P, int WAVES_N, int WAVES_M, int WN_T, int WM_T, int TW, bool M16 = (sizeof(typename P::T) == 2), int NTERM = 1>
static int launch_igemm(const void *in, int n, int H, int W, int cin, int cout, const char *wpk, const float *bias,
                        void *out, int out_split, hipStream_t st, float floor_v = 0.f) {
  typedef ConvGeom<WAVES_N, WAVES_M, WN_T, WM_T, TW> G;
  static std::atomic<bool> attr_done_dev[64];  // the attribute is per device: a process may drive several
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_igemm_kernel<P, WAVES_N, WAVES_M, WN_T, WM_T, TW, M16, NTERM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      set_error("conv3x3_igemm: cannot raise the dynamic LDS limit to %d bytes", G::LDS_BYTES);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int tiles_x = cdiv(W, TW), tiles_y = cdiv(H, G::TH);
  dim3 grid(tiles_x * tiles_y, n, cout / G::BN);
  TimedLaunch t(NQA_K_CONV, st);
  conv3x3_igemm_kernel<P, WAVES_N, WAVES_M, WN_T, WM_T, TW, M16, NTERM><<<grid, G::THREADS, G::LDS_BYTES, st>>>(
      reinterpret_cast<const typename P::T *>(in), wpk, bias, reinterpret_cast<typename P::T *>(out), H, W, cin, cout,
      tiles_x, out_split, floor_v);
  return check_launch("conv3x3_igemm");
}

// compute units of the current device (cached per device: a process may drive several)
static int num_cus();

// conv2_1 (64 -> 128; NCG = 4) and, on two-term weights, conv1_2 (64 -> 64; NCG = 2) with register-resident
// weights, 16-bit kernels; persistent, one 8-wave block per CU
template <typename P, int NCG = 4, int NTERM = 1>
static int launch_regw(const void *in, int n, int H, int W, int layer, const char *packed, void *out, hipStream_t st,
                       int blob_prec = P::ID) {
  constexpr int LDS = 3 * 2 * 1536 * 16;
  static std::atomic<bool> attr_done_dev[64];
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_regw_kernel<P, NCG, NTERM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv3x3_regw: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = num_cus();
  if (!cus) {
    set_error("conv3x3_regw: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 8), total = n * tiles_x * tiles_y;
  const int grid = total < cus ? total : cus;
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, blob_prec));
  TimedLaunch t(NQA_K_CONV, st);
  conv3x3_regw_kernel<P, NCG, NTERM><<<grid, 512, LDS, st>>>(reinterpret_cast<const typename P::T *>(in),
                                                             packed + regw_offset(layer, blob_prec), bias,
                                                             reinterpret_cast<typename P::T *>(out), H, W, tiles_x,
                                                             tiles_y, total);
  return check_launch("conv3x3_regw");
}

// conv2_1 in f32s with register-resident (hi, lo) weights; persistent, one 8-wave block per CU
static int launch_regw_split(const void *in, int n, int H, int W, int layer, const char *packed, void *out, hipStream_t st) {
  constexpr int LDS = 2 * 2 * 204 * 160;  // 130 560
  static std::atomic<bool> attr_done_dev[64];
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_regw_split_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv3x3_regw_split: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = num_cus();
  if (!cus) {
    set_error("conv3x3_regw_split: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int cout = kConvs[layer].cout, nh = cout / 64;
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 4), total = n * tiles_x * tiles_y;
  // the grid is a multiple of 8 XCDs x the layer's 64-channel halves (blocks of one (XCD, stream) share their tiles)
  int streams = cus / (8 * nh);  // 16 on 256 CUs
  if (streams > cdiv(total, 8)) streams = cdiv(total, 8);  // (an XCD owns ~ total / 8 tiles)
  if (streams < 1) streams = 1;
  const int grid = 8 * nh * streams;
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, NQA_PREC_F32S));
  TimedLaunch t(NQA_K_CONV, st);
  conv3x3_regw_split_kernel<<<grid, 512, LDS, st>>>(static_cast<const char *>(in), packed + regw_offset(layer, NQA_PREC_F32S),
                                                    bias, static_cast<char *>(out), H, W, tiles_x, tiles_y, total, cout);
  return check_launch("conv3x3_regw_split");
}

// conv2_2 / conv3_1 (Cin 128) with register-resident weights, 16-bit modes; persistent, one 4-wave block per CU
template <typename P, int NTERM = 1>
static int launch_regw128(const void *in, int n, int H, int W, int layer, const char *packed, void *out, hipStream_t st,
                          int blob_prec = P::ID) {
  constexpr int LDS = 2 * 4 * 1024 * 16, PF = 3;  // fragments three k-steps ahead (2: -6 %, 4: equal)
  static std::atomic<bool> attr_done_dev[64];
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_regw128_kernel<P, PF, NTERM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv3x3_regw128: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = num_cus();
  if (!cus) {
    set_error("conv3x3_regw128: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int cout = kConvs[layer].cout, nct = cout / (128 / NTERM);  // channel tiles (blocks) per pixel tile
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 4), total = n * tiles_x * tiles_y * nct;
  int grid = total < cus ? total : cus;
  // a stride per XCD class that is a multiple of the channel tiles: a block then keeps its channel tile (its weights)
  while (nct > 1 && grid > 8 * nct && ((grid + 7) / 8) % nct) grid -= 8;
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, blob_prec));
  TimedLaunch t(NQA_K_CONV, st);
  conv3x3_regw128_kernel<P, PF, NTERM><<<grid, 256, LDS, st>>>(reinterpret_cast<const typename P::T *>(in),
                                                               packed + regw_offset(layer, blob_prec), bias,
                                                               reinterpret_cast<typename P::T *>(out), H, W, cout,
                                                               tiles_x, tiles_y, total);
  return check_launch("conv3x3_regw128");
}

// two-term weights (f16 kernels on an NQA_PREC_F32M blob): the same tile choices on the NTERM = 2 instances; the
// register-weights kernels hold one-term fragments only, so every layer takes the implicit GEMM here
template <typename P>
static int launch_conv_2term(const void *in, int n, int H, int W, int layer, const char *packed, int blob_prec,
                             const ConvSpec &cs, const char *wpk, const float *bias, void *out, bool narrow, bool big,
                             hipStream_t st) {
  if constexpr (sizeof(typename P::T) == 2 && P::ID == NQA_PREC_F16) {
    // layers 1..4 with the weights in registers, as in the one-term 16-bit modes (first-form bits: the implicit GEMM)
    if (W >= 16 && !g_first_forms) {
      if (layer == 1) return launch_regw<P, 2, 2>(in, n, H, W, layer, packed, out, st, blob_prec);
      if (layer == 2) return launch_regw<P, 4, 2>(in, n, H, W, layer, packed, out, st, blob_prec);
      if ((layer == 3 || layer == 4) && !g_no_regw128) return launch_regw128<P, 2>(in, n, H, W, layer, packed, out, st, blob_prec);
    }
#define NQA_GO2(WN, WM, TN, TM, M16)                                                                                \
  return narrow ? launch_igemm<P, WN, WM, TN, TM, 16, M16, 2>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, 0, st) \
                : launch_igemm<P, WN, WM, TN, TM, 32, M16, 2>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, 0, st)
    if (cs.cout == 64) { NQA_GO2(1, 4, 2, 2, false); }
    if (!big) { NQA_GO2(2, 2, 2, 2, true); }
    NQA_GO2(2, 4, 4, 2, true);
#undef NQA_GO2
  }
  set_error("conv3x3: two-term weights exist for the f16 kernels only");
  return NQA_E_ARG;
}

template <typename P>
static int launch_conv(const void *in, int n, int H, int W, int layer, const char *packed, void *out, hipStream_t st,
                       int blob_prec = P::ID) {
  const ConvSpec &cs = kConvs[layer];
  const char *wpk = packed + layer_offset(layer, blob_prec);
  const float *bias = reinterpret_cast<const float *>(packed + layer_bias_offset(layer, blob_prec));
  const int nterm = layer_terms(blob_prec, layer);
  const bool narrow = W <= 16;  // 32-wide tiles would be half empty
  const int out_split = P::SPLIT && !cs.last;  // f32s: tapped layers leave as float, the others as split16
  // 8-wave 256 ch x 256 px tiles run ~10 % faster per FLOP than 4-wave 128 x 128 tiles on layers
  // with >= 256 output channels (measured), unless their coarser pixel tiling wastes more than most
  // of that on the map's ragged edge (e.g. 68x120) or the map is narrow.  (An 8-wave 128 ch x 256 px
  // tile for the 128-channel layers measured 5-9 % SLOWER than the 4-wave tile.)
  bool big = g_conv_variant >= 1 && cs.cout >= 256 && !narrow;
  if (big) {
    const double eff_big = (double)H * W / ((double)cdiv(W, 32) * cdiv(H, 8) * 256.0);
    const double eff_small = (double)H * W / ((double)cdiv(W, 32) * cdiv(H, 4) * 128.0);
    big = eff_big * 1.05 >= eff_small;  // (1.10 tried: the 68 x 120 maps of a 1080p frame then lose 10 % to the 8-wave grid's tail)
    // small batches: a grid of 8-wave tiles that cannot give every CU a block leaves most of the chip
    // idle for the whole layer; four times as many 4-wave tiles fill it better
    const long blocks_big = (long)cdiv(W, 32) * cdiv(H, 8) * n * (cs.cout / 256);
    if (blocks_big < 192) big = false;
  }
  if (nterm == 2) return launch_conv_2term<P>(in, n, H, W, layer, packed, blob_prec, cs, wpk, bias, out, narrow, big, st);
  // 8-wave 128 ch x 512 px tiles: the loop is bound by what a CU can take in per clock (weights
  // 3*BN*64 B + halo per stage), and for the same 64 K accumulators 128 x 512 moves 37.6 KB per
  // stage where 256 x 256 moves 56 KB and two 128 x 128 blocks move 58 KB.
  bool wide = g_conv_variant >= 2 && cs.cout >= 128 && !narrow && H >= 12;
  if (wide) {
    const double eff_wide = (double)H * W / ((double)cdiv(W, 32) * cdiv(H, 16) * 512.0);
    const double eff_small = (double)H * W / ((double)cdiv(W, 32) * cdiv(H, 4) * 128.0);
    wide = eff_wide * 1.10 >= eff_small;
  }
#define NQA_GO(WN, WM, TN, TM)                                                                                      \
  return narrow ? launch_igemm<P, WN, WM, TN, TM, 16>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, out_split, st) \
                : launch_igemm<P, WN, WM, TN, TM, 32>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, out_split, st)
  if (cs.cout == 64) {  // 64 ch x 256 px, 4 waves; 32x32x16 MFMA: its packed weights are shared with conv1_fused
    return narrow ? launch_igemm<P, 1, 4, 2, 2, 16, false>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, out_split, st)
                  : launch_igemm<P, 1, 4, 2, 2, 32, false>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, out_split, st);
  }
  if constexpr (!P::SPLIT) {
    if (wide) return launch_igemm<P, 2, 4, 2, 4, 32>(in, n, H, W, cs.cin, cs.cout, wpk, bias, out, out_split, st);
  } else {
    // conv2_1 in f32s: register-resident (hi, lo) weights (first-form bit of nqa_set_conv_variant: the implicit GEMM)
    if (layer == 2 && blob_prec == NQA_PREC_F32S && !g_first_forms && W >= 16 && H >= 2)
      return launch_regw_split(in, n, H, W, layer, packed, out, st);
  }
  if constexpr (sizeof(typename P::T) == 2) {
    // conv2_1: register-resident weights (first-form bit of nqa_set_conv_variant: the implicit GEMM, for A/B runs)
    if (blob_prec == P::ID) {  // (their fragments live in the 16-bit blobs only)
      if (layer == 2 && !g_first_forms && W >= 16) return launch_regw<P, 4, 1>(in, n, H, W, layer, packed, out, st);
      if ((layer == 3 || layer == 4) && !g_no_regw128 && !g_first_forms && W >= 16)
        return launch_regw128<P>(in, n, H, W, layer, packed, out, st);
    }
  }
  if (!big) { NQA_GO(2, 2, 2, 2); }                             // 128 ch x 128 px, 4 waves
  NQA_GO(2, 4, 4, 2);                                           // 256 ch x 256 px, 8 waves
#undef NQA_GO
}

// compute units of the current device (cached per device: a process may drive several)
static int num_cus() {
  static std::atomic<int> cus[64];  // (concurrent first calls both query and store the same value)
  const int dev = current_device() & 63;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (!n) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n = prop.multiProcessorCount;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <typename P>
static int launch_conv1_fused(const float *x, const float *y, int B, int n, int H, int W, const char *packed,
                              void *out, hipStream_t st) {
  constexpr int LDS = Conv1Fused::LDS_BYTES;
  static std::atomic<bool> attr_done_dev[64];  // the attribute is per device: a process may drive several
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_fused_kernel<P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv1_fused: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int g_num_cus = num_cus();
  if (!g_num_cus) {
    set_error("conv1_fused: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 4), total = n * tiles_x * tiles_y;
  const int grid = cdiv(total, 2) < g_num_cus ? cdiv(total, 2) : g_num_cus;  // one persistent block per CU
  const char *w1m = packed + layer0_mfma_offset(P::ID);
  const float *b1 = reinterpret_cast<const float *>(packed + layer_bias_offset(0, P::ID));
  const char *w2 = packed + layer_offset(1, P::ID);
  const float *b2 = reinterpret_cast<const float *>(packed + layer_bias_offset(1, P::ID));
  TimedLaunch t(NQA_K_CONV, st);
  conv1_fused_kernel<P><<<grid, 512, LDS, st>>>(x, y, B, w1m, b1, w2, b2, reinterpret_cast<typename P::T *>(out), H, W,
                                                tiles_x, tiles_y, total);
  return check_launch("conv1_fused");
}

template <typename P>
static int launch_conv1_tile(const float *x, const float *y, int B, int n, int H, int W, const char *packed, void *out,
                             hipStream_t st) {
  typedef Conv1Tile::G G;
  constexpr int LDS = Conv1Tile::LDS_BYTES;
  static std::atomic<bool> attr_done_dev[64];  // the attribute is per device: a process may drive several
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_tile_kernel<P>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv1_tile: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, G::TH);
  dim3 grid(tiles_x * tiles_y, n);
  const char *w1m = packed + layer0_mfma_offset(P::ID);
  const float *b1 = reinterpret_cast<const float *>(packed + layer_bias_offset(0, P::ID));
  const char *w2 = packed + layer_offset(1, P::ID);
  const float *b2 = reinterpret_cast<const float *>(packed + layer_bias_offset(1, P::ID));
  TimedLaunch t(NQA_K_CONV, st);
  conv1_tile_kernel<P><<<grid, 256, LDS, st>>>(x, y, B, w1m, b1, w2, b2, reinterpret_cast<typename P::T *>(out), H, W,
                                               tiles_x);
  return check_launch("conv1_tile");
}

template <typename P, int NTERM = 1>
static int launch_conv1_regw(const float *x, const float *y, int B, int n, int H, int W, const char *packed, void *out,
                             hipStream_t st, int blob_prec = P::ID) {
  constexpr int LDS = 2 * 2 * 340 * 96 + 2 * 13 * 40 * 8 + NTERM * 4 * 2 * 64 * 16 + 256 + 3 * 512 * 4;
  static std::atomic<bool> attr_done_dev[64];
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_regw_kernel<P, NTERM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv1_regw: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = num_cus();
  if (!cus) {
    set_error("conv1_regw: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 8), total = n * tiles_x * tiles_y;
  const int grid = total < cus ? total : cus;
  const float *b1 = reinterpret_cast<const float *>(packed + layer_bias_offset(0, blob_prec));
  const float *b2 = reinterpret_cast<const float *>(packed + layer_bias_offset(1, blob_prec));
  // two-term: 1 / conv1_1's weight scale sits in the first float of the (otherwise unused) 32x32 fragment area
  const float *w1inv = reinterpret_cast<const float *>(packed + layer0_mfma_offset(blob_prec));
  TimedLaunch t(NQA_K_CONV, st);
  conv1_regw_kernel<P, NTERM><<<grid, 512, LDS, st>>>(x, y, B, packed + layer0_m16_offset(blob_prec), b1,
                                                      packed + regw_offset(1, blob_prec), b2,
                                                      reinterpret_cast<typename P::T *>(out), H, W, tiles_x, tiles_y,
                                                      total, w1inv);
  return check_launch("conv1_regw");
}

int conv1_fused_split(const float *x, const float *y, int B, int n, int H, int W, const void *packed_v, void *out,
                      hipStream_t st) {
  const char *packed = static_cast<const char *>(packed_v);
  constexpr int LDS = 2 * 2 * 204 * 160 + 2 * 2 * 9 * 40 * 8 + 2 * 4 * 2 * 64 * 16 + 256 + 3 * 320 * 4 + 256;  // 162 816
  static std::atomic<bool> attr_done_dev[64];
  std::atomic<bool> &attr_done = attr_done_dev[current_device() & 63];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_regw_split_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      set_error("conv1_regw_split: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return NQA_E_LAUNCH;
    }
    attr_done = true;
  }
  const int cus = num_cus();
  if (!cus) {
    set_error("conv1_regw_split: cannot query the device");
    return NQA_E_LAUNCH;
  }
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 4), total = n * tiles_x * tiles_y;
  const int grid = total < cus ? total : cus;
  const float *b1 = reinterpret_cast<const float *>(packed + layer_bias_offset(0, NQA_PREC_F32S));
  const float *b2 = reinterpret_cast<const float *>(packed + layer_bias_offset(1, NQA_PREC_F32S));
  const float *w1inv = reinterpret_cast<const float *>(packed + layer0_mfma_offset(NQA_PREC_F32S));
  TimedLaunch t(NQA_K_CONV, st);
  conv1_regw_split_kernel<<<grid, 512, LDS, st>>>(x, y, B, packed + layer0_m16_offset(NQA_PREC_F32S), b1,
                                                  packed + regw_offset(1, NQA_PREC_F32S), b2,
                                                  static_cast<float *>(out), H, W, tiles_x, tiles_y, total, w1inv);
  return check_launch("conv1_regw_split");
}

// stage 1 (conv1_1 + conv1_2) of images [x(0..B), y(0..n-B)) in one kernel; 16-bit modes only
int conv1_fused(const float *x, const float *y, int B, int n, int H, int W, const void *packed, int prec, void *out,
                hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  if (g_stage1_variant == 1) {
    switch (prec) {
      case NQA_PREC_BF16: return launch_conv1_tile<PrecBF16>(x, y, B, n, H, W, p, out, st);
      case NQA_PREC_F16: return launch_conv1_tile<PrecF16>(x, y, B, n, H, W, p, out, st);
    }
  }
  if (g_stage1_variant == 0 && !g_first_forms) {  // the shipped form
    switch (prec) {
      case NQA_PREC_BF16: return launch_conv1_regw<PrecBF16>(x, y, B, n, H, W, p, out, st);
      case NQA_PREC_F16: return launch_conv1_regw<PrecF16>(x, y, B, n, H, W, p, out, st);
    }
  }
  switch (prec) {
    case NQA_PREC_BF16: return launch_conv1_fused<PrecBF16>(x, y, B, n, H, W, p, out, st);
    case NQA_PREC_F16: return launch_conv1_fused<PrecF16>(x, y, B, n, H, W, p, out, st);
  }
  set_error("conv1_fused: 16-bit precision modes only (prec %d)", prec);
  return NQA_E_ARG;
}

int conv1_1(const float *x, int n, int H, int W, const void *packed, int prec, void *out, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  switch (prec) {
    case NQA_PREC_F32S: return launch_conv1_1<PrecF32S>(x, n, H, W, p, out, st);  // split16 out
    case NQA_PREC_F32: return launch_conv1_1<PrecF32>(x, n, H, W, p, out, st);
    case NQA_PREC_BF16: return launch_conv1_1<PrecBF16>(x, n, H, W, p, out, st);
    case NQA_PREC_F16: return launch_conv1_1<PrecF16>(x, n, H, W, p, out, st);
  }
  set_error("conv1_1: unknown prec %d", prec);
  return NQA_E_ARG;
}

int conv3x3_blob(const void *in, int n, int H, int W, int layer, const void *packed, int blob_prec, int kprec,
                 void *out, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  if (is_mixed(blob_prec)) {
    if (kprec == NQA_PREC_F16) return launch_conv<PrecF16>(in, n, H, W, layer, p, out, st, blob_prec);
    if (kprec == NQA_PREC_F32S) return launch_conv<PrecF32S>(in, n, H, W, layer, p, out, st, blob_prec);
  }
  set_error("conv3x3_blob: unsupported blob / kernel precision pair %d / %d", blob_prec, kprec);
  return NQA_E_ARG;
}

// stage 1 of the mixed mode in one kernel (two-term conv1_1 and conv1_2), images [x(0..B), y(0..n-B))
int conv1_fused_blob(const float *x, const float *y, int B, int n, int H, int W, const void *packed, int blob_prec,
                     void *out, hipStream_t st) {
  if (!is_mixed(blob_prec)) {
    set_error("conv1_fused_blob: NQA_PREC_F32M blobs only");
    return NQA_E_ARG;
  }
  return launch_conv1_regw<PrecF16, 2>(x, y, B, n, H, W, static_cast<const char *>(packed), out, st, blob_prec);
}

// A 3x3 convolution that is not a VGG layer of the packed blob: split16 activations in, FLOAT out, weights as one
// f32s-format layer blob (rows + bias + 1/scale, nqa_pack_conv_split), ReLU optional.  Used by the backward pass
// (nqa_backward.hip): the data gradient of a conv layer is the conv of the masked output gradient with the layer's
// flipped, transposed weights -- Cin' = Cout, Cout' = Cin -- without bias or ReLU.
int conv3x3_split_generic(const void *in, int n, int H, int W, int cin, int cout, const void *blob, size_t bias_off,
                          int relu, void *out, hipStream_t st) {
  const char *wpk = static_cast<const char *>(blob);
  const float *bias = reinterpret_cast<const float *>(wpk + bias_off);
  const float fl = relu ? 0.f : -INFINITY;
  const bool narrow = W <= 16;
  if (cout % 128 == 0) {
    return narrow ? launch_igemm<PrecF32S, 2, 2, 2, 2, 16>(in, n, H, W, cin, cout, wpk, bias, out, 0, st, fl)
                  : launch_igemm<PrecF32S, 2, 2, 2, 2, 32>(in, n, H, W, cin, cout, wpk, bias, out, 0, st, fl);
  }
  return narrow ? launch_igemm<PrecF32S, 1, 4, 2, 2, 16, false>(in, n, H, W, cin, cout, wpk, bias, out, 0, st, fl)
                : launch_igemm<PrecF32S, 1, 4, 2, 2, 32, false>(in, n, H, W, cin, cout, wpk, bias, out, 0, st, fl);
}

int conv1_1_blob(const float *x, int n, int H, int W, const void *packed, int blob_prec, int kprec, void *out,
                 hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  if (is_mixed(blob_prec) && kprec == NQA_PREC_F16) return launch_conv1_1<PrecF16>(x, n, H, W, p, out, st, blob_prec);
  set_error("conv1_1_blob: unsupported blob / kernel precision pair %d / %d", blob_prec, kprec);
  return NQA_E_ARG;
}

int conv3x3(const void *in, int n, int H, int W, int layer, const void *packed, int prec, void *out, hipStream_t st) {
  const char *p = static_cast<const char *>(packed);
  switch (prec) {
    case NQA_PREC_F32: return launch_conv<PrecF32>(in, n, H, W, layer, p, out, st);
    case NQA_PREC_BF16: return launch_conv<PrecBF16>(in, n, H, W, layer, p, out, st);
    case NQA_PREC_F16: return launch_conv<PrecF16>(in, n, H, W, layer, p, out, st);
    case NQA_PREC_F32S: return launch_conv<PrecF32S>(in, n, H, W, layer, p, out, st);
  }
  set_error("conv3x3: unknown prec %d", prec);
  return NQA_E_ARG;
}

}  // namespace nqa
