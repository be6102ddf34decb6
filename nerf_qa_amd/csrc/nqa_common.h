// Shared device/host helpers for libnqa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nqa.h"

namespace nqa {

// ---- error plumbing ------------------------------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);

// ---- timing ring (bench.py roofline leg) -----------------------------------------
struct TimedLaunch {
  TimedLaunch(int kclass, hipStream_t s);
  ~TimedLaunch();
  int kclass;
  hipStream_t stream;
  int slot;
};

// ---- precision traits ------------------------------------------------------------
// One 16-byte "k-chunk" is the unit every tile is staged and read in: 8 sixteen-bit
// channels or 4 floats.  A 64-byte pixel record (4 chunks) is one LDS row.
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// T: element type of the maps a kernel READS (and, normally, writes); TO: element type of what the pooling
// kernels WRITE (differs from T only at the mixed mode's f16 -> split16 boundary, PrecF16X).
struct PrecF32 {
  typedef float T;
  typedef float TO;
  static constexpr bool OUT_SPLIT16 = false;
  static constexpr int ID = NQA_PREC_F32;
  static constexpr int CPC = 4;   // channels per 16-byte chunk
  static constexpr int KC = 16;   // channels per 64-byte LDS row
  static constexpr bool SPLIT = false;
  __device__ static inline float to_f(T v) { return v; }
  __device__ static inline T from_f(float v) { return v; }
  // one chunk pair -> K=8 of the contraction as four exact-f32 MFMAs (K=2 each)
  __device__ static inline f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
};
// NQA_PREC_F32S: float precision carried as f16 (hi, lo) pairs so the convolutions run on the f16
// MFMA: a*b ~= ah*bh + ah*bl + al*bh (the dropped al*bl is ~2^-22 relative).  Weights are split
// at pack time.  Activations BETWEEN conv layers live in the "split16" format, written by the
// producing kernel's epilogue: 16 channels = 64 bytes [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15],
// hi = f16(v), lo = f16(v - hi) -- 4 bytes per element like float, same row layout as the packed
// weights, so both MFMA operands are plain 16-byte LDS reads.  The tapped maps (relu1_2 ... relu5_3,
// read by the statistics / pooling / A-DISTS kernels) are plain float.
struct PrecF32S : PrecF32 {
  static constexpr int ID = NQA_PREC_F32S;
  static constexpr bool SPLIT = true;
};
struct PrecBF16 {
  typedef __bf16 T;
  typedef __bf16 TO;
  static constexpr bool OUT_SPLIT16 = false;
  static constexpr int ID = NQA_PREC_BF16;
  static constexpr int CPC = 8;
  static constexpr int KC = 32;
  static constexpr bool SPLIT = false;
  __device__ static inline float to_f(T v) { return (float)v; }
  __device__ static inline T from_f(float v) { return (T)v; }
  __device__ static inline f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
  }
};
struct PrecF16 {
  typedef _Float16 T;
  typedef _Float16 TO;
  static constexpr bool OUT_SPLIT16 = false;
  static constexpr int ID = NQA_PREC_F16;
  static constexpr int CPC = 8;
  static constexpr int KC = 32;
  static constexpr bool SPLIT = false;
  __device__ static inline float to_f(T v) { return (float)v; }
  __device__ static inline T from_f(float v) { return (T)v; }
  __device__ static inline f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
  }
};

// NQA_PREC_F32M, the L2-pool behind its last two-term stage: half taps in, split16 records out (they feed an
// f32s conv layer)
struct PrecF16X : PrecF16 {
  typedef float TO;  // 4 bytes per element
  static constexpr bool OUT_SPLIT16 = true;
};

// split16 store of the 4 consecutive channels c..c+3 (c % 4 == 0) of the pixel record at `pixel`
__device__ static inline void store_split4(char *pixel, int c, float v0, float v1, float v2, float v3) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const h4 hi = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
  const h4 lo = {(_Float16)(v0 - (float)hi[0]), (_Float16)(v1 - (float)hi[1]), (_Float16)(v2 - (float)hi[2]),
                 (_Float16)(v3 - (float)hi[3])};
  char *p = pixel + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
  *reinterpret_cast<h4 *>(p) = hi;
  *reinterpret_cast<h4 *>(p + 32) = lo;
}
// store one P::CPC-channel group (channels c..) of a pixel in the activation format `P` writes
template <typename P>
__device__ static inline void store_group(typename P::T *pixel, int c, const float (&v)[P::CPC]) {
  if constexpr (P::SPLIT) {
    store_split4(reinterpret_cast<char *>(pixel), c, v[0], v[1], v[2], v[3]);
  } else {
    typedef __attribute__((ext_vector_type(P::CPC))) typename P::T tvec;
    tvec o;
#pragma unroll
    for (int e = 0; e < P::CPC; ++e) o[e] = P::from_f(v[e]);
    *reinterpret_cast<tvec *>(pixel + c) = o;
  }
}

// what the pooling kernels write: P's own format, or split16 records at the mixed mode's boundary
template <typename P>
__device__ static inline void store_pooled(typename P::TO *pixel, int c, const float (&v)[P::CPC]) {
  if constexpr (P::OUT_SPLIT16 && sizeof(typename P::T) == 2) {
    store_split4(reinterpret_cast<char *>(pixel), c, v[0], v[1], v[2], v[3]);
    store_split4(reinterpret_cast<char *>(pixel), c + 4, v[4], v[5], v[6], v[7]);
  } else {
    store_group<P>(reinterpret_cast<typename P::T *>(pixel), c, v);
  }
}

// (NQA_PREC_F32M is a pyramid-level mode: its stages run as F16 / F32S kernels, see stage_prec; the packed blob
// holds 4 bytes per weight in it -- f16 hi + lo, or the f32s rows)
__host__ __device__ static inline size_t prec_elem_bytes(int prec) {
  return prec == NQA_PREC_F32 || prec == NQA_PREC_F32S ? 4 : 2;
}
static inline bool prec_valid(int prec) { return prec >= NQA_PREC_F32 && prec <= NQA_PREC_F32S; }  // kernel-level modes
static inline bool is_mixed(int prec) { return NQA_MIXED_STAGES(prec) > 0; }
static inline bool prec_valid_pyramid(int prec) { return prec_valid(prec) || is_mixed(prec); }
// the kernel precision of pyramid stage `stage` (0-based) in mode `prec`
static inline int stage_prec(int prec, int stage) {
  if (!is_mixed(prec)) return prec;
  return stage < NQA_MIXED_STAGES(prec) ? NQA_PREC_F16 : NQA_PREC_F32S;
}
// weight terms per product of conv layer `layer` in mode `prec` (2: f16 hi + lo against f16 activations)
static inline int layer_terms(int prec, int layer);
// the precision the kernels that READ tapped maps see: in f32s those are plain float
static inline int storage_prec(int prec) { return prec == NQA_PREC_F32S ? NQA_PREC_F32 : prec; }

// ---- VGG plan --------------------------------------------------------------------
struct ConvSpec {
  int cin, cout, stage;  // stage 0..4 (after which tap k=stage+1 is taken when last==1)
  int last;
};
static const ConvSpec kConvs[NQA_NUM_CONVS] = {
    {3, 64, 0, 0},    {64, 64, 0, 1},   {64, 128, 1, 0},  {128, 128, 1, 1}, {128, 256, 2, 0},
    {256, 256, 2, 0}, {256, 256, 2, 1}, {256, 512, 3, 0}, {512, 512, 3, 0}, {512, 512, 3, 1},
    {512, 512, 4, 0}, {512, 512, 4, 0}, {512, 512, 4, 1}};
static inline int layer_terms(int prec, int layer) {
  return kConvs[layer].stage < NQA_MIXED_STAGES(prec) ? 2 : 1;
}
static const int kChns[NQA_NUM_TAPS] = {3, 64, 128, 256, 512, 512};
static const int kChnOff[NQA_NUM_TAPS] = {0, 3, 67, 195, 451, 963};

// Packed-blob layout (bytes).  [0,256): zero page (source of out-of-image halo pixels).
// Layer 0: float w[27][64] (k = (ky*3+kx)*3+c) then float bias[64].  Layers 1..12: tiles
// [cout/64][cin/KC][9 taps][64 rows] of 64 bytes (chunk c of row n stored at position
// c ^ ((n>>2)&3), or c ^ 2*((n>>2)&1) for the 16-bit layers 2..12 read by the 16x16x32 MFMA), then
// float bias[cout].  The 64-channel granularity lets any block tile
// that is a multiple of 64 channels stream whole sub-slabs.
static constexpr size_t kZeroPage = 256;
size_t layer_offset(int layer, int prec);
size_t layer_bias_offset(int layer, int prec);
// conv1_1 again as 16-bit MFMA A fragments [ky][cout][h][8], k = ky*16 + kx*4 + c (zero for
// kx = 3 or c = 3), 6144 bytes, for the fused stage-1 kernel (zeros in the f32 blob)
size_t layer0_mfma_offset(int prec);
// 16x16x32 MFMA A fragments of a 64-input-channel layer (1 or 2), 16-bit modes: see nqa_api.hip
size_t regw_offset(int layer, int prec);
size_t layer0_m16_offset(int prec);  // conv1_1 as 16x16x32 A fragments (16-bit modes)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Where each stage's partial sums live and how to fold them (finalize_kernel).
struct StageDesc {
  long part_off[NQA_NUM_TAPS];  // offset (in doubles) of each stage's partial block
  int nblk[NQA_NUM_TAPS];
  int hw[NQA_NUM_TAPS];
  int c[NQA_NUM_TAPS];
  int coff[NQA_NUM_TAPS];  // channel offset of the stage inside the concatenated vector
  int nstage;
  int ctot;
};

size_t max_act_elems(int H, int W);  // largest activation map of the pyramid, elements per image (nqa_api.hip)

// ---- host launchers shared between translation units ---------------------------------
void set_conv_variant(int v);
void set_conv_first_forms(int on);
void set_adists_window_legacy(bool on);
int conv1_1(const float *x, int n, int H, int W, const void *packed, int prec, void *out, hipStream_t st);
int conv1_fused(const float *x, const float *y, int B, int n, int H, int W, const void *packed, int prec, void *out,
                hipStream_t st);
int conv3x3(const void *in, int n, int H, int W, int layer, const void *packed, int prec, void *out, hipStream_t st);
// one conv layer of a blob packed for `blob_prec` run by the kernels of `kprec` (mixed mode: F16 with two-term
// weights for layers 1..6, F32S for the rest); out_float: an F32S layer leaves plain float (a tapped map)
int conv3x3_blob(const void *in, int n, int H, int W, int layer, const void *packed, int blob_prec, int kprec,
                 void *out, hipStream_t st);
int conv1_1_blob(const float *x, int n, int H, int W, const void *packed, int blob_prec, int kprec, void *out,
                 hipStream_t st);
int conv1_fused_blob(const float *x, const float *y, int B, int n, int H, int W, const void *packed, int blob_prec,
                     void *out, hipStream_t st);
bool mixed_stage1_unfused();  // (A/B switch of nqa_set_conv_variant's first-forms bit)
// stage 1 of an f32s blob in one kernel (three-term products, float NHWC out), images [x(0..B), y(0..n-B))
int conv1_fused_split(const float *x, const float *y, int B, int n, int H, int W, const void *packed, void *out,
                      hipStream_t st);
int relu_mask_split16(const float *g, const void *act, int act_split, long npix, int C, void *out, hipStream_t st);
int l2pool_backward(const float *x, const void *y_split16, const float *gy, int n, int H, int W, int C, float *gx,
                    hipStream_t st);
int conv1_1_backward(const float *gm, const float *w_oihw, int n, int H, int W, float *gimg, hipStream_t st);
int conv3x3_split_generic(const void *in, int n, int H, int W, int cin, int cout, const void *blob, size_t bias_off,
                          int relu, void *out, hipStream_t st);
int l2pool_to_split16(const void *in_f16, int n, int H, int W, int C, void *out_split16, hipStream_t st);
int pool_stats_to_split16(const void *feat_f16, int B, int H, int W, int C, void *pooled_split16, double *part,
                          hipStream_t st);
int l2pool(const void *in, int n, int H, int W, int C, int prec, void *out, hipStream_t st);
int stats_units_per_block(int units, int C, int prec, int B);
int stats_nchw_ppb(int HW);
int pool_stats_tiles(int Ho, int Wo, int C, int prec, int B, int *tr, int *tc);
int pool_stats(const void *feat, int B, int H, int W, int C, int prec, void *pooled, double *part, hipStream_t st);
int stats_nhwc(const void *feat, int B, int HW, int C, int prec, double *part, hipStream_t st);
int stats_nchw(const float *fx, const float *fy, int B, int C, int HW, double *part, hipStream_t st);
int finalize(const double *part, const StageDesc &d, int B, float *s1, float *s2, hipStream_t st);
int score(const float *s1, const float *s2, const float *alpha, const float *beta, int B, float *out, hipStream_t st);
int nhwc_to_nchw(const void *in, int n, int HW, int C, int prec, float *out, hipStream_t st);
// ---- conv + L2-pool + statistics in one kernel (nqa_conv_pool.hip; the DISTS path's tap 2) ----
// rows of statistics partials a fused tap reserves per pair (= the largest grid the fused kernel runs; unwritten rows are zero)
#define NQA_FUSED_PART_BLOCKS 256
#define NQA_FUSED_PART_BLOCKS_S1 512  // (the fused stage 1 leaves two rows per block)
void set_fuse_taps(int on);
void set_fuse_stage1(int on);
int pool_seam_finish(const float *seam, void *pooled, int nimg, int strips, int Ho, int Wo, int C, hipStream_t st);
bool conv1_pool_fusable(int B, int H, int W, int blob_prec);
int conv1_pool_stats_fused(const float *x, const float *y, int B, int H, int W, const void *packed, void *pooled, float *seam,
                           double *part, hipStream_t st);
bool conv_pool_fusable(int layer, int B, int H, int W, int blob_prec, int kprec);
size_t conv_pool_seam_bytes(int B, int H, int W, int C);
int conv_pool_stats_fused(const void *in, int B, int H, int W, int layer, const void *packed, int blob_prec, void *pooled,
                          float *seam, double *part, hipStream_t st);

}  // namespace nqa
