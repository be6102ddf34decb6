// extern "C" surface of libnqa_hip.so (include/nqa.h): error plumbing, timing ring,
// weight packing, and the drivers that chain the kernels into forward_once /
// DISTS.forward (nerf_qa/DISTS_pytorch/DISTS_pt.py:91-148).
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <thread>
#include <vector>

#include "nqa_common.h"

namespace nqa {

// ---- errors ---------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NQA_E_LAUNCH;
  }
  return NQA_OK;
}

// ---- timing ring ------------------------------------------------------------------
// Thread-local like the error string: a thread that enables timing brackets ITS OWN launches (on whatever
// streams it uses) and collects only those, so concurrent callers on other threads / streams are neither
// timed nor disturbed.  The events are created once per thread and reused.
static const int kRing = 16384;
static thread_local bool g_timing = false;
static thread_local std::vector<hipEvent_t> g_ev0, g_ev1;
static thread_local std::vector<int> g_cls;
static thread_local int g_used = 0;
// The events are created lazily, up to the high-water mark a thread actually used, and given back when the thread
// ends -- except on the thread that loaded the library (normally the process's main thread): its thread_locals are
// destroyed during process exit, when the HIP runtime may already be gone.
static const std::thread::id g_load_thread = std::this_thread::get_id();
struct TimingRingOwner {
  ~TimingRingOwner() {
    if (std::this_thread::get_id() == g_load_thread) return;
    for (hipEvent_t e : g_ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : g_ev1) (void)hipEventDestroy(e);
    g_ev0.clear();
    g_ev1.clear();
  }
};
static thread_local TimingRingOwner g_ring_owner;  // constructed after the vectors (first touched below), destroyed before them

TimedLaunch::TimedLaunch(int kc, hipStream_t s) : kclass(kc), stream(s), slot(-1) {
  if (!g_timing || g_used >= kRing) return;
  if ((int)g_ev0.size() <= g_used) {
    (void)&g_ring_owner;  // (odr-use: the owner exists in every thread that ever creates events)
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_ev0.push_back(a);
    g_ev1.push_back(b);
    g_cls.push_back(kc);
  }
  slot = g_used++;
  g_cls[slot] = kc;
  (void)hipEventRecord(g_ev0[slot], stream);
}
TimedLaunch::~TimedLaunch() {
  if (slot >= 0) (void)hipEventRecord(g_ev1[slot], stream);
}

// ---- packed weight blob -------------------------------------------------------------
// bytes per weight in the blob of `prec`: NQA_PREC_F32M holds f16 (hi, lo) pairs for layers 1..6 and f32s rows
// (also hi + lo halves) for layers 7..12
static size_t blob_weight_bytes(int prec) { return is_mixed(prec) ? 4 : prec_elem_bytes(prec); }
static size_t layer_bytes(int layer, int prec) {
  const ConvSpec &c = kConvs[layer];
  if (layer == 0) return align_up(27 * 64 * 4 + 64 * 4, 256) + 6144;
  // weights, then float bias[cout] followed by ONE float: 1 / (the layer's power-of-two weight scale)
  return align_up((size_t)c.cin * c.cout * 9 * blob_weight_bytes(prec), 256) + align_up((size_t)c.cout * 4 + 4, 256);
}
size_t layer_offset(int layer, int prec) {
  size_t o = kZeroPage;
  for (int l = 0; l < layer; ++l) o += layer_bytes(l, prec);
  return o;
}
size_t layer0_mfma_offset(int prec) { return layer_offset(0, prec) + align_up(27 * 64 * 4 + 64 * 4, 256); }
// Register-resident weight fragments of layers 1..4 (conv1_2, conv2_1: Cin 64; conv2_2, conv3_1: Cin 128), 16-bit
// modes only: the whole layer as 16x16x32 MFMA A fragments [cout/32][2 tiles of 16][Cin/32*9 k-steps][64 lanes][8 halfs],
// k-step = chunk * 9 + tap (chunk = 32 input channels), lane = (row l15, k-group c4): element j is
// w[cout = 32*g + 16*i + l15][cin = 32*chunk + 8*c4 + j][tap].  conv3x3_regw_kernel loads its 2 x 18 fragments
// (144 VGPRs) once per persistent block.  Appended behind the ordinary layers.
// (NQA_PREC_F32M: the same fragments for 16 channels per group, once as the f16 `hi` and once as the `lo` part of the
// scaled weights: [cout/16][part][k-steps][64 lanes][8 halfs])
// (NQA_PREC_F32S: conv1_2 and conv2_1 in the two-term form, for conv1_regw_split_kernel and conv3x3_regw_split_kernel)
size_t regw_bytes(int layer, int prec) {
  if (prec == NQA_PREC_F32S) return layer <= 2 ? (size_t)kConvs[layer].cout * 64 * 9 * 2 * 2 : 0;  // conv1_2, conv2_1
  return (size_t)kConvs[layer].cout * kConvs[layer].cin * 9 * 2 * (is_mixed(prec) ? 2 : 1);
}
static const int kRegwFirst = 1, kRegwLast = 4;  // conv1_2, conv2_1 (Cin 64); conv2_2, conv3_1 (Cin 128)
// conv1_1 as 16x16x32 MFMA A fragments for conv1_regw_kernel: [4 tiles of 16 channels][2 MFMAs][64 lanes][8 halfs];
// MFMA m contracts kernel rows 2m and 2m+1: k = 16*(ky - 2m) + 4*kx + c (kx, c padded to 4; zero for ky = 3)
static constexpr size_t kW1M16Bytes = 4 * 2 * 64 * 16;
size_t regw_offset(int layer, int prec) {
  size_t o = layer_offset(NQA_NUM_CONVS, prec);
  for (int l = kRegwFirst; l < layer; ++l) o += regw_bytes(l, prec);
  return o;
}
size_t layer0_m16_offset(int prec) { return regw_offset(kRegwLast + 1, prec); }
size_t layer_bias_offset(int layer, int prec) {
  const ConvSpec &c = kConvs[layer];
  if (layer == 0) return layer_offset(0, prec) + 27 * 64 * 4;
  return layer_offset(layer, prec) + align_up((size_t)c.cin * c.cout * 9 * blob_weight_bytes(prec), 256);
}

static uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);                    // round to nearest even
}
static uint16_t f32_to_f16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7FFFFFFFu;
  if (u >= 0x7F800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));
  if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to >= 65520 -> inf
  if (u < 0x33000001u) return (uint16_t)sign;               // < 2^-25 (or == 2^-25 tie -> 0)
  const int e = (int)(u >> 23) - 127;
  uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) {  // subnormal half
    shift = 13 + (-14 - e);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7FFFFFu;
  }
  uint32_t r = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (r & 1u))) ++r;
  return (uint16_t)(sign | (base + r));
}
static float f16_to_f32(uint16_t hv) {
  const int e = (hv >> 10) & 31, m = hv & 1023;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf((float)(m | 1024), e - 25);
  return (hv & 0x8000) ? -v : v;
}

// power of two that puts a layer's largest |w| in [512, 1024): the f16 `lo` part of a scaled weight is then a normal
// half (see the f32s packing below); the conv epilogues multiply by the exact inverse
static int weight_scale_exp(const float *w, size_t count) {
  float wmax = 0.f;
  for (size_t i = 0; i < count; ++i) wmax = fmaxf(wmax, fabsf(w[i]));
  int k = 0;
  if (wmax > 0.f && isfinite(wmax)) {
    k = (int)floor(log2(1024.0 / (double)wmax));
    k = k < -8 ? -8 : (k > 24 ? 24 : k);
  }
  return k;
}

// ---- drivers ----------------------------------------------------------------------------
struct PyrDims {
  int h[5], w[5];
};
static PyrDims pyr_dims(int H, int W) {
  PyrDims d;
  d.h[0] = H;
  d.w[0] = W;
  for (int k = 1; k < 5; ++k) {
    d.h[k] = (d.h[k - 1] + 1) / 2;
    d.w[k] = (d.w[k - 1] + 1) / 2;
  }
  return d;
}

// Elements of the largest activation map of the pyramid, per image.  For ordinary frames that is
// stage 1 (H*W*64), but the maps shrink by ceil(./2) while the channels double, so below ~8 pixels
// a side a LATER stage is the largest (a 1x1 frame: 64 elements at stage 1, 512 at stages 4 and 5);
// sizing the ping-pong buffers by stage 1 alone let those stages write past them.
size_t max_act_elems(int H, int W) {
  static const int kStageC[5] = {64, 128, 256, 512, 512};
  const PyrDims d = pyr_dims(H, W);
  size_t m = 0;
  for (int k = 0; k < 5; ++k) {
    const size_t e = (size_t)d.h[k] * d.w[k] * kStageC[k];
    m = e > m ? e : m;
  }
  return m;
}
static size_t act_bytes(int n, int H, int W, int prec) {
  if (is_mixed(prec)) {  // the largest map in BYTES: stages differ in element size
    static const int kStageC[5] = {64, 128, 256, 512, 512};
    const PyrDims d = pyr_dims(H, W);
    size_t m = 0;
    for (int k = 0; k < 5; ++k) {
      const size_t e = (size_t)d.h[k] * d.w[k] * kStageC[k] * prec_elem_bytes(stage_prec(prec, k));
      m = e > m ? e : m;
      if (k > 0) {  // stage k's input: stage k-1's pooled map in stage k's format
        const size_t p = (size_t)d.h[k] * d.w[k] * kStageC[k - 1] * prec_elem_bytes(stage_prec(prec, k));
        m = p > m ? p : m;
      }
    }
    return align_up((size_t)n * m, 256);
  }
  return align_up((size_t)n * max_act_elems(H, W) * prec_elem_bytes(prec), 256);
}

// Runs the 13 convs and the four L2-pools on `n` images: images [0,nx) come from x, the rest from
// y (fp32 NCHW).  Stage 1 is the fused conv1_1+conv1_2 kernel in the 16-bit modes and two kernels
// in f32.  Stage k's last conv writes into taps[k] when taps is given, else into the ping-pong pair.
// on_tap(k, tap, Hk, Wk, Ck, pool_dst) is called once that conv is enqueued; pool_dst is where
// the stage's L2-pool output must go (null after stage 5).  It returns 1 if it pooled the tap
// itself (the fused pool+statistics pass), 0 to have the plain L2-pool run, <0 on error.
// fuse_tap(k, in, Hk, Wk, layer, pool_dst): the DISTS path's offer to run stage k's LAST conv, its L2-pool and its
// statistics as one kernel (nqa_conv_pool.hip); it returns 1 if it did, 0 to decline (then the conv, on_tap and the
// pool run as usual), < 0 on error.  Only asked when the batch is x | y pairs (n == 2 * nx) and no taps are wanted.
struct NoFuse {
  int operator()(int, const void *, int, int, int, void *) const { return 0; }
};
// fuse_stage1(pool_dst): the same offer for the whole of stage 1 (nqa_conv1_pool.hip: normalisation, conv1_1, conv1_2,
// L2-pool and the statistics of tap 1 from the raw images); 1 = done, `pool_dst` holds the pooled relu1_2.
struct NoFuse1 {
  int operator()(void *) const { return 0; }
};
template <typename F, typename FU = NoFuse, typename FS = NoFuse1>
static int run_stages(const float *x, const float *y, int nx, void *bufA, void *bufB, int n, int H, int W,
                      const void *packed, int prec, void *const *taps, F on_tap, hipStream_t st, FU fuse_tap = FU(),
                      FS fuse_stage1 = FS()) {
  const PyrDims d = pyr_dims(H, W);
  void *cur = bufA;
  int rc;
  if (is_mixed(prec)) {
    // mixed mode: conv1_1 exact in float -> half; layers 1..6 the f16 kernels on two-term weights; the pool behind
    // stage 3 writes split16 records; layers 7..12 the f32s kernels.  on_tap sees the stage's own kernel precision.
    const bool fused_m = W >= 16 && !mixed_stage1_unfused();
    if (!fused_m) {  // conv1_1 exact in float -> half, then conv1_2 as a layer of its own
      if ((rc = conv1_1_blob(x, nx, H, W, packed, prec, NQA_PREC_F16, bufA, st))) return rc;
      if (n > nx && (rc = conv1_1_blob(y, n - nx, H, W, packed, prec, NQA_PREC_F16,
                                       static_cast<char *>(bufA) + (size_t)nx * H * W * 64 * 2, st)))
        return rc;
    }
    for (int layer = 1; layer < NQA_NUM_CONVS; ++layer) {
      const ConvSpec &cs = kConvs[layer];
      const int k = cs.stage, kp = stage_prec(prec, k);
      void *dst = (cs.last && taps) ? taps[k] : (cur == bufA ? bufB : bufA);
      if (cs.last && k < 4 && !taps && n == 2 * nx && layer > 1) {
        if ((rc = fuse_tap(k, cur, d.h[k], d.w[k], layer, dst)) < 0) return rc;
        if (rc == 1) {  // conv + pool + statistics done: `dst` holds the POOLED map
          cur = dst;
          continue;
        }
      }
      if (layer == 1 && fused_m) {
        if ((rc = conv1_fused_blob(x, y, nx, n, H, W, packed, prec, dst, st))) return rc;
      } else if ((rc = conv3x3_blob(cur, n, d.h[k], d.w[k], layer, packed, prec, kp, dst, st))) {
        return rc;
      }
      cur = dst;
      if (cs.last) {
        void *pdst = k < 4 ? ((cur == bufA) ? bufB : bufA) : nullptr;
        if ((rc = on_tap(k, cur, d.h[k], d.w[k], cs.cout, pdst)) < 0) return rc;
        if (k < 4) {
          if (rc == 0) {
            const bool boundary = kp == NQA_PREC_F16 && stage_prec(prec, k + 1) == NQA_PREC_F32S;
            rc = boundary ? l2pool_to_split16(cur, n, d.h[k], d.w[k], cs.cout, pdst, st)
                          : l2pool(cur, n, d.h[k], d.w[k], cs.cout, kp, pdst, st);
            if (rc) return rc;
          }
          cur = pdst;
        }
      }
    }
    return NQA_OK;
  }
  // f32s: conv1_regw_split_kernel (three-term products, relu1_1 stays in LDS) unless the first-forms bit asks for the
  // round-2 pair conv1_1_kernel (VALU, split16 out) + implicit GEMM
  const bool fused_s = prec == NQA_PREC_F32S && !mixed_stage1_unfused();
  const bool fused1 = prec_elem_bytes(prec) == 2 || fused_s;
  int first_layer = 1;
  if (!taps && n == 2 * nx) {  // stage 1 with its pool and statistics in one kernel: the layer loop starts at conv2_1
    if ((rc = fuse_stage1(bufA)) < 0) return rc;
    if (rc == 1) first_layer = 2;
  }
  if (!fused1 && first_layer == 1) {
    if ((rc = conv1_1(x, nx, H, W, packed, prec, bufA, st))) return rc;
    if (n > nx && (rc = conv1_1(y, n - nx, H, W, packed, prec,
                                static_cast<char *>(bufA) + (size_t)nx * H * W * 64 * prec_elem_bytes(prec), st)))
      return rc;
  }
  for (int layer = first_layer; layer < NQA_NUM_CONVS; ++layer) {
    const ConvSpec &cs = kConvs[layer];
    const int k = cs.stage;
    void *dst = (cs.last && taps) ? taps[k] : (cur == bufA ? bufB : bufA);
    if (cs.last && k < 4 && !taps && n == 2 * nx && layer > 1) {
      if ((rc = fuse_tap(k, cur, d.h[k], d.w[k], layer, dst)) < 0) return rc;
      if (rc == 1) {
        cur = dst;
        continue;
      }
    }
    if (layer == 1 && fused1) {
      if ((rc = fused_s ? conv1_fused_split(x, y, nx, n, H, W, packed, dst, st)
                        : conv1_fused(x, y, nx, n, H, W, packed, prec, dst, st)))
        return rc;
    } else if ((rc = conv3x3(cur, n, d.h[k], d.w[k], layer, packed, prec, dst, st))) {
      return rc;
    }
    cur = dst;
    if (cs.last) {
      void *pdst = k < 4 ? ((cur == bufA) ? bufB : bufA) : nullptr;
      if ((rc = on_tap(k, cur, d.h[k], d.w[k], cs.cout, pdst)) < 0) return rc;
      if (k < 4) {
        if (rc == 0 && (rc = l2pool(cur, n, d.h[k], d.w[k], cs.cout, prec, pdst, st))) return rc;
        cur = pdst;
      }
    }
  }
  return NQA_OK;
}

// `chan`: the widest H x W map the call addresses with 32-bit in-image byte offsets (the conv DMA
// plan): 64 channels for the pyramid paths (H, W are stage-1 sizes; later stages have 1/4 of the
// pixels per doubling of the channels), the layer's own input width for the single-operator calls.
static bool bad_dims(const char *who, int n, int H, int W, int prec, int chan = 512, bool pyramid = false) {
  if (n <= 0 || H <= 0 || W <= 0) {
    set_error("%s: non-positive size n=%d H=%d W=%d", who, n, H, W);
    return true;
  }
  if (!(pyramid ? prec_valid_pyramid(prec) : prec_valid(prec))) {
    set_error(is_mixed(prec) ? "%s: NQA_PREC_F32M / F32M2 (%d) is a mode of the DISTS pyramid entry points only"
                                    : "%s: unknown prec %d", who, prec);
    return true;
  }
  if (chan > 0 && (long)H * W * chan * (long)prec_elem_bytes(prec) >= (1L << 31)) {
    set_error("%s: map too large for 32-bit in-image byte offsets (H*W*%d channels*%d bytes >= 2^31)", who, chan,
              (int)prec_elem_bytes(prec));
    return true;
  }
  return false;
}

struct StatsPlan {
  StageDesc d;
  size_t doubles;
};
// `pooled[k]` = {Ho, Wo} where stage k's statistics come out of the fused pool+statistics pass
// ({0,0}: the plain NHWC pass over HW[k]); pooled == null: every stage runs the NCHW plane kernel.
static StatsPlan stats_plan(int B, const int *C, const int *HW, const int (*pooled)[2], int nstage, int prec,
                            const bool *fused = nullptr) {
  StatsPlan p;
  memset(&p, 0, sizeof(p));
  long off = 0;
  int coff = 0;
  for (int k = 0; k < nstage; ++k) {
    const bool nchw = !pooled || k == 0;
    const int kp = k >= 1 ? stage_prec(prec, k - 1) : prec;  // tap k comes out of pyramid stage k-1 (0-based)
    int nblk;
    if (nchw)
      nblk = cdiv(HW[k], stats_nchw_ppb(HW[k]));
    else if (fused && fused[k])  // (rows per pair reserved by the fused conv + pool + statistics kernels; unwritten rows are zero)
      nblk = k == 1 ? NQA_FUSED_PART_BLOCKS_S1 : NQA_FUSED_PART_BLOCKS;
    else if (pooled[k][0])
      nblk = pool_stats_tiles(pooled[k][0], pooled[k][1], C[k], kp, B, nullptr, nullptr);
    else
      nblk = cdiv(HW[k], stats_units_per_block(HW[k], C[k], kp, B));
    p.d.part_off[k] = off;
    p.d.nblk[k] = nblk;
    p.d.hw[k] = HW[k];
    p.d.c[k] = C[k];
    p.d.coff[k] = coff;
    off += (long)B * p.d.nblk[k] * C[k] * 5;
    coff += C[k];
  }
  p.d.nstage = nstage;
  p.d.ctot = coff;
  p.doubles = (size_t)off;
  return p;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_version(void) { return NQA_VERSION; }
const char *nqa_last_error(void) { return g_err; }

int nqa_set_conv_variant(int variant) {
  if (variant < 0 || variant > 255 || (variant & 3) == 3) {
    set_error("set_conv_variant: unknown variant %d", variant);
    return NQA_E_ARG;
  }
  set_conv_variant(variant & 7);
  set_adists_window_legacy((variant & 8) != 0);
  set_conv_first_forms((variant >> 4) & 3);
  set_fuse_taps((variant & 64) ? 0 : 1);
  set_fuse_stage1((variant & 128) ? 0 : 1);
  return NQA_OK;
}

int nqa_timing_enable(int on) {
  g_timing = on != 0;
  g_used = 0;
  return NQA_OK;
}

int nqa_timing_collect(int launches[NQA_K_COUNT], double ms[NQA_K_COUNT]) {
  for (int i = 0; i < NQA_K_COUNT; ++i) {
    launches[i] = 0;
    ms[i] = 0.0;
  }
  for (int s = 0; s < g_used; ++s) {
    float t = 0.f;
    if (hipEventSynchronize(g_ev1[s]) != hipSuccess || hipEventElapsedTime(&t, g_ev0[s], g_ev1[s]) != hipSuccess) {
      set_error("timing_collect: event %d failed", s);
      g_used = 0;
      return NQA_E_LAUNCH;
    }
    launches[g_cls[s]] += 1;
    ms[g_cls[s]] += (double)t;
  }
  g_used = 0;
  return NQA_OK;
}

size_t nqa_packed_weights_bytes(int prec) {
  if (!prec_valid_pyramid(prec)) return 0;
  size_t n = layer_offset(NQA_NUM_CONVS, prec);
  if (is_mixed(prec) || prec_elem_bytes(prec) == 2 || prec == NQA_PREC_F32S)
    n = layer0_m16_offset(prec) + kW1M16Bytes * (is_mixed(prec) || prec == NQA_PREC_F32S ? 2 : 1);
  return n;
}

int nqa_pack_vgg_weights(const float *const w_host[NQA_NUM_CONVS], const float *const b_host[NQA_NUM_CONVS], int prec,
                         void *packed_host) {
  if (!w_host || !b_host || !packed_host) {
    set_error("pack_vgg_weights: null pointer");
    return NQA_E_ARG;
  }
  if (!prec_valid_pyramid(prec)) {
    set_error("pack_vgg_weights: unknown prec %d", prec);
    return NQA_E_ARG;
  }
  char *blob = static_cast<char *>(packed_host);
  memset(blob, 0, nqa_packed_weights_bytes(prec));
  {  // layer 0: w0[k][co], k = (ky*3+kx)*3 + c, from OIHW
    float *w0 = reinterpret_cast<float *>(blob + layer_offset(0, prec));
    for (int co = 0; co < 64; ++co)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 9; ++t) w0[(t * 3 + c) * 64 + co] = w_host[0][(co * 3 + c) * 9 + t];
    memcpy(blob + layer_bias_offset(0, prec), b_host[0], 64 * 4);
    if (!is_mixed(prec) && prec_elem_bytes(prec) == 2) {  // MFMA A fragments for the fused stage-1 kernel
      uint16_t *wm = reinterpret_cast<uint16_t *>(blob + layer0_mfma_offset(prec));
      for (int ky = 0; ky < 3; ++ky)
        for (int co = 0; co < 64; ++co)
          for (int hh = 0; hh < 2; ++hh)
            for (int j = 0; j < 8; ++j) {
              const int kx = 2 * hh + j / 4, c = j % 4;
              const float v = (kx < 3 && c < 3) ? w_host[0][(co * 3 + c) * 9 + ky * 3 + kx] : 0.f;
              wm[((ky * 64 + co) * 2 + hh) * 8 + j] = prec == NQA_PREC_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
            }
    }
  }
  for (int l = 1; l < NQA_NUM_CONVS; ++l) {
    const ConvSpec &cs = kConvs[l];
    // the kernel precision that reads this layer: NQA_PREC_F32M = f16 kernels (two-term weights) for layers 1..6,
    // f32s kernels for the rest
    const int lprec = stage_prec(prec, cs.stage), nterm = layer_terms(prec, l);
    const int cpc = prec_elem_bytes(lprec) == 4 ? 4 : 8, kc = 4 * cpc;
    const int bn = 64, ncc = cs.cin / kc;
    char *dst = blob + layer_offset(l, prec);
    {  // 1 / weight scale behind the bias (1 in every mode but f32s)
      float one = 1.f;
      memcpy(blob + layer_bias_offset(l, prec) + (size_t)cs.cout * 4, &one, 4);
    }
    if (nterm == 2) {
      // f16 (hi, lo) pairs of the weights times a power of two (as for f32s below: lo must be a NORMAL half), in
      // the stage order of conv3x3_igemm_kernel<..., NTERM = 2>: per 64-channel sub-slab and 32-channel chunk,
      // [ky][part: hi, lo][kx][64 rows][4 chunks of 8 halfs]; chunk swizzle as in the one-term 16-bit blob
      const int k = weight_scale_exp(w_host[l], (size_t)cs.cout * cs.cin * 9);
      const float wscale = ldexpf(1.f, k), winv = ldexpf(1.f, -k);
      memcpy(blob + layer_bias_offset(l, prec) + (size_t)cs.cout * 4, &winv, 4);
      for (int ct = 0; ct < cs.cout / bn; ++ct)
        for (int cc = 0; cc < ncc; ++cc)
          for (int ky = 0; ky < 3; ++ky)
            for (int part = 0; part < 2; ++part)
              for (int kx = 0; kx < 3; ++kx)
                for (int n = 0; n < bn; ++n)
                  for (int pos = 0; pos < 4; ++pos) {
                    const int c = pos ^ (l >= 2 ? ((n >> 2) & 1) * 2 : (n >> 2) & 3);
                    uint16_t *row = reinterpret_cast<uint16_t *>(dst) +
                                    ((((((((size_t)ct * ncc + cc) * 3 + ky) * 2 + part) * 3 + kx) * bn + n) * 4 + pos) * 8);
                    for (int j = 0; j < 8; ++j) {
                      const int cin = cc * 32 + c * 8 + j, cout = ct * bn + n;
                      const float v = w_host[l][((size_t)cout * cs.cin + cin) * 9 + ky * 3 + kx] * wscale;
                      const uint16_t hi = f32_to_f16(v);
                      row[j] = part == 0 ? hi : f32_to_f16(v - f16_to_f32(hi));
                    }
                  }
      memcpy(blob + layer_bias_offset(l, prec), b_host[l], (size_t)cs.cout * 4);
      continue;
    }
    if (lprec == NQA_PREC_F32S) {
      // rows of 16 input channels as [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] halves, lo = f16(w*s - hi).
      // s is a power of two per layer that puts the layer's largest |w| in [512, 1024): a VGG weight is
      // ~1e-2, whose residual after the f16 `hi` (~5e-6) is a SUBNORMAL half with an absolute quantum of 6e-8
      // -- 25 times coarser than float32 relative to the weight, which left 2e-6 of noise on every
      // pre-activation and flipped nearly-dead channels of A-DISTS (oracle/knife_edge_study.py).  Scaled,
      // hi and lo are both normal halves (22-23 significant bits together); the scale is exact and the
      // conv epilogue multiplies the accumulator by 1/s (also exact) before the bias.
      float wmax = 0.f;
      for (size_t i = 0; i < (size_t)cs.cout * cs.cin * 9; ++i) wmax = fmaxf(wmax, fabsf(w_host[l][i]));
      int k = 0;
      if (wmax > 0.f && isfinite(wmax)) {
        k = (int)floor(log2(1024.0 / (double)wmax));
        k = k < -8 ? -8 : (k > 24 ? 24 : k);
      }
      const float wscale = ldexpf(1.f, k), winv = ldexpf(1.f, -k);
      memcpy(blob + layer_bias_offset(l, prec) + (size_t)cs.cout * 4, &winv, 4);
      for (int ct = 0; ct < cs.cout / bn; ++ct)
        for (int cc = 0; cc < ncc; ++cc)
          for (int t = 0; t < 9; ++t)
            for (int n = 0; n < bn; ++n)
              for (int pos = 0; pos < 4; ++pos) {
                const int c = pos ^ ((n >> 2) & 3);
                uint16_t *row = reinterpret_cast<uint16_t *>(dst) + (((((size_t)ct * ncc + cc) * 9 + t) * bn + n) * 4 + pos) * 8;
                for (int j = 0; j < 8; ++j) {
                  const int cin = cc * 16 + (c & 1) * 8 + j, cout = ct * bn + n;
                  const float v = w_host[l][((size_t)cout * cs.cin + cin) * 9 + t] * wscale;
                  const uint16_t hi = f32_to_f16(v);
                  row[j] = c < 2 ? hi : f32_to_f16(v - f16_to_f32(hi));
                }
              }
      memcpy(blob + layer_bias_offset(l, prec), b_host[l], (size_t)cs.cout * 4);
      continue;
    }
    for (int ct = 0; ct < cs.cout / bn; ++ct)
      for (int cc = 0; cc < ncc; ++cc)
        for (int t = 0; t < 9; ++t)
          for (int n = 0; n < bn; ++n)
            for (int pos = 0; pos < 4; ++pos) {
              // chunk swizzle of the MFMA shape that reads the layer: 16x16x32 for the 16-bit igemm
              // layers (2..13), 32x32x16 for conv1_2 (shared with the fused stage-1 kernel) and f32
              const int c = pos ^ ((cpc == 8 && l >= 2) ? ((n >> 2) & 1) * 2 : (n >> 2) & 3);
              const size_t e0 = (((((size_t)ct * ncc + cc) * 9 + t) * bn + n) * 4 + pos) * cpc;
              for (int j = 0; j < cpc; ++j) {
                const int cin = cc * kc + c * cpc + j, cout = ct * bn + n;
                const float v = w_host[l][((size_t)cout * cs.cin + cin) * 9 + t];
                if (lprec == NQA_PREC_F32)
                  reinterpret_cast<float *>(dst)[e0 + j] = v;
                else
                  reinterpret_cast<uint16_t *>(dst)[e0 + j] = lprec == NQA_PREC_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
              }
            }
    memcpy(blob + layer_bias_offset(l, prec), b_host[l], (size_t)cs.cout * 4);
  }
  if (is_mixed(prec) || prec == NQA_PREC_F32S) {
    // conv1_1 for conv1_regw_kernel<.., NTERM = 2> (and, in f32s, conv1_regw_split_kernel): [4 tiles of 16 channels][2
    // MFMAs][part: hi, lo][64 lanes][8 halfs] of
    // the weights times a power of two; 1 / that scale in the first float of the 32x32-fragment area
    const int k1 = weight_scale_exp(w_host[0], 64 * 27);
    const float s1 = ldexpf(1.f, k1), inv1 = ldexpf(1.f, -k1);
    memcpy(blob + layer0_mfma_offset(prec), &inv1, 4);
    uint16_t *w1 = reinterpret_cast<uint16_t *>(blob + layer0_m16_offset(prec));
    for (int i = 0; i < 4; ++i)
      for (int m = 0; m < 2; ++m)
        for (int part = 0; part < 2; ++part)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int k = 8 * (lane >> 4) + j, ky = 2 * m + (k >> 4), kx = (k >> 2) & 3, c = k & 3;
              const int cout = 16 * i + (lane & 15);
              const float v = (ky < 3 && kx < 3 && c < 3) ? w_host[0][(cout * 3 + c) * 9 + ky * 3 + kx] * s1 : 0.f;
              const uint16_t hi = f32_to_f16(v);
              w1[((((size_t)i * 2 + m) * 2 + part) * 64 + lane) * 8 + j] = part == 0 ? hi : f32_to_f16(v - f16_to_f32(hi));
            }
    // two-term register fragments of layers 1..4 (conv3x3_regw_kernel<.., NTERM = 2> and regw128; f32s: conv1_2 only)
    for (int l = kRegwFirst; l <= (prec == NQA_PREC_F32S ? kRegwFirst + 1 : kRegwLast); ++l) {
      const ConvSpec &cs = kConvs[l];
      const int nks = cs.cin / 32 * 9;
      const float wscale = ldexpf(1.f, weight_scale_exp(w_host[l], (size_t)cs.cout * cs.cin * 9));  // (as the layer's rows)
      uint16_t *dst = reinterpret_cast<uint16_t *>(blob + regw_offset(l, prec));
      for (int g = 0; g < cs.cout / 16; ++g)
        for (int part = 0; part < 2; ++part)
          for (int ks = 0; ks < nks; ++ks)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int cout = 16 * g + (lane & 15), cin = 32 * (ks / 9) + 8 * (lane >> 4) + j, t = ks % 9;
                const float v = w_host[l][((size_t)cout * cs.cin + cin) * 9 + t] * wscale;
                const uint16_t hi = f32_to_f16(v);
                dst[((((size_t)g * 2 + part) * nks + ks) * 64 + lane) * 8 + j] = part == 0 ? hi : f32_to_f16(v - f16_to_f32(hi));
              }
    }
  }
  if (!is_mixed(prec) && prec_elem_bytes(prec) == 2) {
    for (int l = kRegwFirst; l <= kRegwLast; ++l) {
      const ConvSpec &cs = kConvs[l];
      const int nks = cs.cin / 32 * 9;  // k-steps: chunk * 9 + tap
      uint16_t *dst = reinterpret_cast<uint16_t *>(blob + regw_offset(l, prec));
      for (int g = 0; g < cs.cout / 32; ++g)
        for (int i = 0; i < 2; ++i)
          for (int ks = 0; ks < nks; ++ks)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int cout = 32 * g + 16 * i + (lane & 15), cin = 32 * (ks / 9) + 8 * (lane >> 4) + j, t = ks % 9;
                const float v = w_host[l][((size_t)cout * cs.cin + cin) * 9 + t];
                dst[((((size_t)g * 2 + i) * nks + ks) * 64 + lane) * 8 + j] =
                    prec == NQA_PREC_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
              }
    }
    uint16_t *w1 = reinterpret_cast<uint16_t *>(blob + layer0_m16_offset(prec));
    for (int i = 0; i < 4; ++i)
      for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int k = 8 * (lane >> 4) + j, ky = 2 * m + (k >> 4), kx = (k >> 2) & 3, c = k & 3;
            const int cout = 16 * i + (lane & 15);
            const float v = (ky < 3 && kx < 3 && c < 3) ? w_host[0][(cout * 3 + c) * 9 + ky * 3 + kx] : 0.f;
            w1[(((size_t)i * 2 + m) * 64 + lane) * 8 + j] = prec == NQA_PREC_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
          }
  }
  return NQA_OK;
}

int nqa_conv1_1(const float *x, int n, int H, int W, const void *packed, int prec, void *out, void *stream) {
  if (!x || !packed || !out) {
    set_error("conv1_1: null pointer");
    return NQA_E_ARG;
  }
  if (bad_dims("conv1_1", n, H, W, prec, 64)) return NQA_E_ARG;
  return conv1_1(x, n, H, W, packed, prec, out, static_cast<hipStream_t>(stream));
}

int nqa_conv1_fused(const float *x, int n, int H, int W, const void *packed, int prec, void *out, void *stream) {
  if (!x || !packed || !out) {
    set_error("conv1_fused: null pointer");
    return NQA_E_ARG;
  }
  if (bad_dims("conv1_fused", n, H, W, prec, 64)) return NQA_E_ARG;
  if (prec == NQA_PREC_F32S) return conv1_fused_split(x, nullptr, n, n, H, W, packed, out, static_cast<hipStream_t>(stream));
  return conv1_fused(x, nullptr, n, n, H, W, packed, prec, out, static_cast<hipStream_t>(stream));
}

int nqa_conv3x3_relu(const void *in, int n, int H, int W, int layer, const void *packed, int prec, void *out,
                     void *stream) {
  if (!in || !packed || !out) {
    set_error("conv3x3_relu: null pointer");
    return NQA_E_ARG;
  }
  if (layer < 1 || layer >= NQA_NUM_CONVS) {
    set_error("conv3x3_relu: layer %d out of range 1..12", layer);
    return NQA_E_ARG;
  }
  if (bad_dims("conv3x3_relu", n, H, W, prec, kConvs[layer].cin > kConvs[layer].cout ? kConvs[layer].cin : kConvs[layer].cout))
    return NQA_E_ARG;
  return conv3x3(in, n, H, W, layer, packed, prec, out, static_cast<hipStream_t>(stream));
}

int nqa_l2pool(const void *in, int n, int H, int W, int C, int prec, void *out, void *stream) {
  if (!in || !out) {
    set_error("l2pool: null pointer");
    return NQA_E_ARG;
  }
  if (bad_dims("l2pool", n, H, W, prec, C > 0 ? C : 1)) return NQA_E_ARG;
  if (C <= 0 || C % 8) {
    set_error("l2pool: C=%d must be a positive multiple of 8", C);
    return NQA_E_SHAPE;
  }
  return l2pool(in, n, H, W, C, prec, out, static_cast<hipStream_t>(stream));
}

int nqa_nhwc_to_nchw_f32(const void *in, int n, int H, int W, int C, int prec, float *out, void *stream) {
  if (!in || !out) {
    set_error("nhwc_to_nchw: null pointer");
    return NQA_E_ARG;
  }
  // the export kernel indexes with size_t: no 32-bit in-image offset bound here (a 1080p float tap is 531 MB
  // per image at 64 channels and must pass); only the grid's limits apply
  if (bad_dims("nhwc_to_nchw", n, H, W, prec, 0) || C <= 0) return NQA_E_ARG;
  if ((long)H * W > (1L << 31) - 64 || n > 65535 || C > 65535 * 64) {  // grid = (HW/64, C/64, n)
    set_error("nhwc_to_nchw: map too large for the launch grid (H*W=%ld, C=%d, n=%d)", (long)H * W, C, n);
    return NQA_E_SHAPE;
  }
  return nhwc_to_nchw(in, n, H * W, C, prec, out, static_cast<hipStream_t>(stream));
}

// Statistics plan of the fused DISTS path: stage 0 from the raw images (NCHW kernel), taps 1..4
// inside the fused pool+statistics pass (items = pooled pixels), tap 5 by the plain NHWC pass.
// which taps (index 1..5 of the statistics plan) the DISTS path runs fused with their conv (nqa_conv_pool.hip)
static void dists_fused_taps(int B, int H, int W, int prec, bool fused[6]) {
  const PyrDims d = pyr_dims(H, W);
  for (int k = 0; k < 6; ++k) fused[k] = false;
  fused[1] = conv1_pool_fusable(B, H, W, prec);
  fused[2] = conv_pool_fusable(3, B, d.h[1], d.w[1], prec, stage_prec(prec, 1));
}
static StatsPlan dists_stats_plan(int B, int H, int W, int prec, bool allow_fused = true) {
  const PyrDims d = pyr_dims(H, W);
  int C[6], HW[6], pooled[6][2];
  bool fused[6];
  dists_fused_taps(B, H, W, prec, fused);
  if (!allow_fused)
    for (int k = 0; k < 6; ++k) fused[k] = false;
  C[0] = 3;
  HW[0] = H * W;
  pooled[0][0] = pooled[0][1] = 0;
  for (int k = 0; k < 5; ++k) {
    C[k + 1] = kChns[k + 1];
    HW[k + 1] = d.h[k] * d.w[k];
    pooled[k + 1][0] = k < 4 ? d.h[k + 1] : 0;
    pooled[k + 1][1] = k < 4 ? d.w[k + 1] : 0;
  }
  return stats_plan(B, C, HW, pooled, 6, prec, fused);
}
// seam planes of the fused taps (behind the statistics partials in the workspace)
static size_t dists_seam_bytes(int B, int H, int W) {  // (one area: tap 1's planes are consumed before tap 2's are written)
  const PyrDims d = pyr_dims(H, W);
  const size_t s1 = conv_pool_seam_bytes(B, H, W, 64), s2 = conv_pool_seam_bytes(B, d.h[1], d.w[1], 128);
  return s1 > s2 ? s1 : s2;
}

size_t nqa_workspace_bytes(int n_images, int H, int W, int prec) {
  if (n_images <= 0 || H <= 0 || W <= 0) return 0;
  // (sized for the fused and for the unfused form of every tap: the choice can be switched per thread, nqa_set_conv_variant)
  const int B = (n_images + 1) / 2;
  const size_t pa = dists_stats_plan(B, H, W, prec, true).doubles, pb = dists_stats_plan(B, H, W, prec, false).doubles;
  return 2 * act_bytes(n_images, H, W, prec) + align_up((pa > pb ? pa : pb) * 8, 256) + dists_seam_bytes(B, H, W);
}

int nqa_vgg_pyramid(const float *x, int n, int H, int W, const void *packed, int prec, void *ws, size_t ws_bytes,
                    void *const taps[5], void *stream) {
  if (!x || !packed || !ws || !taps) {
    set_error("vgg_pyramid: null pointer");
    return NQA_E_ARG;
  }
  for (int k = 0; k < 5; ++k)
    if (!taps[k]) {
      set_error("vgg_pyramid: taps[%d] is null", k);
      return NQA_E_ARG;
    }
  if (bad_dims("vgg_pyramid", n, H, W, prec, 64, true)) return NQA_E_ARG;
  const size_t ab = act_bytes(n, H, W, prec);
  if (ws_bytes < 2 * ab) {
    set_error("vgg_pyramid: workspace %zu < %zu bytes", ws_bytes, 2 * ab);
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *bufA = static_cast<char *>(ws), *bufB = bufA + ab;
  return run_stages(x, nullptr, n, bufA, bufB, n, H, W, packed, prec, taps,
                    [](int, void *, int, int, int, void *) { return 0; }, st);
}

int nqa_dists_forward(const float *x, const float *y, int B, int H, int W, const void *packed, int prec, void *ws,
                      size_t ws_bytes, float *s1, float *s2, void *stream) {
  if (!x || !y || !packed || !ws || !s1 || !s2) {
    set_error("dists_forward: null pointer");
    return NQA_E_ARG;
  }
  if (bad_dims("dists_forward", B, H, W, prec, 64, true)) return NQA_E_ARG;
  const int n = 2 * B;
  const size_t need = nqa_workspace_bytes(n, H, W, prec);
  if (ws_bytes < need) {
    set_error("dists_forward: workspace %zu < %zu bytes", ws_bytes, need);
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t ab = act_bytes(n, H, W, prec);
  char *bufA = static_cast<char *>(ws), *bufB = bufA + ab;
  double *part = reinterpret_cast<double *>(bufB + ab);
  const StatsPlan p = dists_stats_plan(B, H, W, prec);
  const size_t pa = p.doubles, pb = dists_stats_plan(B, H, W, prec, false).doubles;
  float *seam = reinterpret_cast<float *>(reinterpret_cast<char *>(part) + align_up((pa > pb ? pa : pb) * 8, 256));
  bool fused[6];
  dists_fused_taps(B, H, W, prec, fused);
  int rc;
  // feature 0 is the raw image (DISTS_pt.py:103): statistics straight from the inputs
  if ((rc = stats_nchw(x, y, B, 3, H * W, part + p.d.part_off[0], st))) return rc;
  // x images occupy [0,B), y images [B,2B) of one NHWC batch
  rc = run_stages(
      x, y, B, bufA, bufB, n, H, W, packed, prec, nullptr,
      [&](int k, void *tap, int hk, int wk, int ck, void *pool_dst) {
        double *pk = part + p.d.part_off[k + 1];
        const int kp = stage_prec(prec, k);  // (mixed mode: half taps up to stage 3, float ones behind)
        if (!pool_dst) return stats_nhwc(tap, B, hk * wk, ck, kp, pk, st);
        const bool boundary = is_mixed(prec) && kp == NQA_PREC_F16 && stage_prec(prec, k + 1) == NQA_PREC_F32S;
        const int rc2 = boundary ? pool_stats_to_split16(tap, B, hk, wk, ck, pool_dst, pk, st)
                                 : pool_stats(tap, B, hk, wk, ck, kp, pool_dst, pk, st);
        return rc2 ? rc2 : 1;
      },
      st,
      [&](int k, const void *inp, int hk, int wk, int layer, void *pool_dst) {
        if (!fused[k + 1]) return 0;
        const int rc2 = conv_pool_stats_fused(inp, B, hk, wk, layer, packed, prec, pool_dst, seam, part + p.d.part_off[k + 1], st);
        return rc2 ? rc2 : 1;
      },
      [&](void *pool_dst) {
        if (!fused[1]) return 0;
        const int rc2 = conv1_pool_stats_fused(x, y, B, H, W, packed, pool_dst, seam, part + p.d.part_off[1], st);
        return rc2 ? rc2 : 1;
      });
  if (rc) return rc;
  return finalize(part, p.d, B, s1, s2, st);
}

size_t nqa_stats_scratch_bytes(int B, const int C[NQA_NUM_TAPS], const int Hk[NQA_NUM_TAPS],
                               const int Wk[NQA_NUM_TAPS]) {
  if (B <= 0 || !C || !Hk || !Wk) return 0;
  int HW[6];
  for (int k = 0; k < 6; ++k) HW[k] = Hk[k] * Wk[k];
  return align_up(stats_plan(B, C, HW, nullptr, 6, NQA_PREC_F32).doubles * 8, 256);
}

int nqa_dists_stats_nchw(const float *const fx[NQA_NUM_TAPS], const float *const fy[NQA_NUM_TAPS], int B,
                         const int C[NQA_NUM_TAPS], const int Hk[NQA_NUM_TAPS], const int Wk[NQA_NUM_TAPS],
                         void *scratch, size_t scratch_bytes, float *s1, float *s2, void *stream) {
  if (!fx || !fy || !C || !Hk || !Wk || !scratch || !s1 || !s2 || B <= 0) {
    set_error("dists_stats_nchw: bad argument");
    return NQA_E_ARG;
  }
  int HW[6];
  for (int k = 0; k < 6; ++k) {
    if (!fx[k] || !fy[k] || C[k] <= 0 || Hk[k] <= 0 || Wk[k] <= 0) {
      set_error("dists_stats_nchw: bad feature %d", k);
      return NQA_E_ARG;
    }
    HW[k] = Hk[k] * Wk[k];
  }
  const StatsPlan p = stats_plan(B, C, HW, nullptr, 6, NQA_PREC_F32);
  if (scratch_bytes < p.doubles * 8) {
    set_error("dists_stats_nchw: scratch %zu < %zu bytes", scratch_bytes, p.doubles * 8);
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *part = static_cast<double *>(scratch);
  int rc;
  for (int k = 0; k < 6; ++k)
    if ((rc = stats_nchw(fx[k], fy[k], B, C[k], HW[k], part + p.d.part_off[k], st))) return rc;
  return finalize(part, p.d, B, s1, s2, st);
}

// ---- conv + L2-pool + statistics as a single operator (tests, tools; the DISTS path calls the launcher directly) ----
static __global__ void part_reduce_kernel(const double *__restrict__ part, int nblk, int C, double *__restrict__ sums, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (b, c, s)
  if (idx >= total) return;
  const long per = (long)C * 5;
  const long b = idx / per, cs = idx - b * per;
  double acc = 0.0;
  for (int k = 0; k < nblk; ++k) acc += part[((size_t)b * nblk + k) * per + cs];
  sums[idx] = acc;
}

static int fused_part_rows(int layer) { return layer == 1 ? NQA_FUSED_PART_BLOCKS_S1 : NQA_FUSED_PART_BLOCKS; }
int nqa_dists_fused_taps(int B, int H, int W, int prec, int fused[6]) {
  if (!fused) return NQA_E_ARG;
  if (bad_dims("dists_fused_taps", B, H, W, prec, 0, true)) return NQA_E_ARG;
  bool f[6];
  dists_fused_taps(B, H, W, prec, f);
  for (int k = 0; k < 6; ++k) fused[k] = f[k] ? 1 : 0;
  return NQA_OK;
}
size_t nqa_conv_pool_workspace_bytes(int B, int H, int W, int layer) {
  if (B <= 0 || H <= 0 || W <= 0 || layer < 1 || layer >= NQA_NUM_CONVS) return 0;
  const int C = kConvs[layer].cout;
  return align_up((size_t)B * fused_part_rows(layer) * C * 5 * sizeof(double), 256) + conv_pool_seam_bytes(B, H, W, C);
}

int nqa_conv1_pool_stats(const float *x, const float *y, int B, int H, int W, const void *packed, int prec, void *pooled,
                         double *sums, void *ws, size_t ws_bytes, void *stream) {
  if (!x || !y || !packed || !pooled || !sums || !ws || B <= 0 || H <= 0 || W <= 0) {
    set_error("conv1_pool_stats: bad argument");
    return NQA_E_ARG;
  }
  if (!conv1_pool_fusable(B, H, W, prec)) {
    set_error("conv1_pool_stats: no fused stage 1 for %d x %d frames in precision %d (B = %d)", H, W, prec, B);
    return NQA_E_SHAPE;
  }
  if (ws_bytes < nqa_conv_pool_workspace_bytes(B, H, W, 1)) {
    set_error("conv1_pool_stats: workspace %zu < %zu bytes", ws_bytes, nqa_conv_pool_workspace_bytes(B, H, W, 1));
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *part = static_cast<double *>(ws);
  float *seam = reinterpret_cast<float *>(static_cast<char *>(ws) +
                                          align_up((size_t)B * NQA_FUSED_PART_BLOCKS_S1 * 64 * 5 * sizeof(double), 256));
  const int rc = conv1_pool_stats_fused(x, y, B, H, W, packed, pooled, seam, part, st);
  if (rc) return rc;
  const long total = (long)B * 64 * 5;
  part_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, NQA_FUSED_PART_BLOCKS_S1, 64, sums, total);
  return check_launch("part_reduce");
}

int nqa_conv_pool_stats(const void *in, int B, int H, int W, int layer, const void *packed, int prec, void *pooled,
                        double *sums, void *ws, size_t ws_bytes, void *stream) {
  if (!in || !packed || !pooled || !sums || !ws) {
    set_error("conv_pool_stats: null pointer");
    return NQA_E_ARG;
  }
  if (layer < 1 || layer >= NQA_NUM_CONVS || B <= 0 || H <= 0 || W <= 0 || !prec_valid_pyramid(prec)) {
    set_error("conv_pool_stats: bad argument (layer %d, B %d, %d x %d, prec %d)", layer, B, H, W, prec);
    return NQA_E_ARG;
  }
  const int kp = stage_prec(prec, kConvs[layer].stage);
  if (!conv_pool_fusable(layer, B, H, W, prec, kp)) {
    set_error("conv_pool_stats: no fused form for layer %d of a %d x %d map in precision %d (B = %d)", layer, H, W, prec, B);
    return NQA_E_SHAPE;
  }
  if (ws_bytes < nqa_conv_pool_workspace_bytes(B, H, W, layer)) {
    set_error("conv_pool_stats: workspace %zu < %zu bytes", ws_bytes, nqa_conv_pool_workspace_bytes(B, H, W, layer));
    return NQA_E_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int C = kConvs[layer].cout;
  double *part = static_cast<double *>(ws);
  float *seam = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)B * NQA_FUSED_PART_BLOCKS * C * 5 * sizeof(double), 256));
  const int rc = conv_pool_stats_fused(in, B, H, W, layer, packed, prec, pooled, seam, part, st);
  if (rc) return rc;
  const long total = (long)B * C * 5;
  part_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(part, NQA_FUSED_PART_BLOCKS, C, sums, total);
  return check_launch("part_reduce");
}

int nqa_dists_score(const float *s1, const float *s2, const float *alpha, const float *beta, int B, float *out,
                    void *stream) {
  if (!s1 || !s2 || !alpha || !beta || !out || B <= 0) {
    set_error("dists_score: bad argument");
    return NQA_E_ARG;
  }
  return score(s1, s2, alpha, beta, B, out, static_cast<hipStream_t>(stream));
}


// ---- backward pass (DISTS require_grad=True; nqa_backward.hip) -----------------------------------------------------
static size_t conv_split_bias_off(int cout, int cin) { return align_up((size_t)cout * cin * 9 * 4, 256); }

size_t nqa_packed_conv_split_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cout % 64 || cin % 16) return 0;
  return conv_split_bias_off(cout, cin) + align_up((size_t)cout * 4 + 4, 256);
}

int nqa_pack_conv_split(const float *w_oihw, int cout, int cin, void *packed_host) {
  if (!w_oihw || !packed_host || cout <= 0 || cin <= 0 || cout % 64 || cin % 16) {
    set_error("pack_conv_split: bad argument (cout %d must be a multiple of 64, cin %d of 16)", cout, cin);
    return NQA_E_ARG;
  }
  char *blob = static_cast<char *>(packed_host);
  memset(blob, 0, nqa_packed_conv_split_bytes(cout, cin));  // (bias = 0)
  const int k = weight_scale_exp(w_oihw, (size_t)cout * cin * 9);
  const float wscale = ldexpf(1.f, k), winv = ldexpf(1.f, -k);
  memcpy(blob + conv_split_bias_off(cout, cin) + (size_t)cout * 4, &winv, 4);
  const int bn = 64, ncc = cin / 16;  // the f32s row layout of nqa_pack_vgg_weights
  for (int ct = 0; ct < cout / bn; ++ct)
    for (int cc = 0; cc < ncc; ++cc)
      for (int t = 0; t < 9; ++t)
        for (int n = 0; n < bn; ++n)
          for (int pos = 0; pos < 4; ++pos) {
            const int c = pos ^ ((n >> 2) & 3);
            uint16_t *row = reinterpret_cast<uint16_t *>(blob) + (((((size_t)ct * ncc + cc) * 9 + t) * bn + n) * 4 + pos) * 8;
            for (int j = 0; j < 8; ++j) {
              const int ci = cc * 16 + (c & 1) * 8 + j, co = ct * bn + n;
              const float v = w_oihw[((size_t)co * cin + ci) * 9 + t] * wscale;
              const uint16_t hi = f32_to_f16(v);
              row[j] = c < 2 ? hi : f32_to_f16(v - f16_to_f32(hi));
            }
          }
  return NQA_OK;
}

int nqa_conv3x3_split(const void *in_split16, int n, int H, int W, int cin, int cout, const void *packed_conv, int relu,
                      float *out_nhwc, void *stream) {
  if (!in_split16 || !packed_conv || !out_nhwc) {
    set_error("conv3x3_split: null pointer");
    return NQA_E_ARG;
  }
  if (cout <= 0 || cin <= 0 || cout % 64 || cin % 16) {
    set_error("conv3x3_split: cout %d must be a multiple of 64, cin %d of 16", cout, cin);
    return NQA_E_SHAPE;
  }
  if (bad_dims("conv3x3_split", n, H, W, NQA_PREC_F32S, cin > cout ? cin : cout)) return NQA_E_ARG;
  return conv3x3_split_generic(in_split16, n, H, W, cin, cout, packed_conv, conv_split_bias_off(cout, cin), relu,
                               out_nhwc, static_cast<hipStream_t>(stream));
}

int nqa_relu_mask_split16(const float *g_nhwc, const void *act_nhwc, int act_is_split16, long pixels, int C,
                          void *out_split16, void *stream) {
  if (!g_nhwc || !act_nhwc || !out_split16 || pixels <= 0 || C <= 0 || C % 16) {
    set_error("relu_mask_split16: bad argument (C %d must be a positive multiple of 16)", C);
    return NQA_E_ARG;
  }
  return relu_mask_split16(g_nhwc, act_nhwc, act_is_split16, pixels, C, out_split16, static_cast<hipStream_t>(stream));
}

int nqa_l2pool_backward(const float *tap_nhwc, const void *pooled_split16, const float *g_pooled_nhwc, int n, int H, int W,
                        int C, float *g_tap_nhwc, void *stream) {
  if (!tap_nhwc || !pooled_split16 || !g_pooled_nhwc || !g_tap_nhwc || C <= 0 || C % 16) {
    set_error("l2pool_backward: bad argument");
    return NQA_E_ARG;
  }
  if (bad_dims("l2pool_backward", n, H, W, NQA_PREC_F32, 0)) return NQA_E_ARG;
  return l2pool_backward(tap_nhwc, pooled_split16, g_pooled_nhwc, n, H, W, C, g_tap_nhwc, static_cast<hipStream_t>(stream));
}

int nqa_conv1_1_backward(const float *gm_nhwc, const float *w_oihw_dev, int n, int H, int W, float *g_image_nchw,
                         void *stream) {
  if (!gm_nhwc || !w_oihw_dev || !g_image_nchw) {
    set_error("conv1_1_backward: null pointer");
    return NQA_E_ARG;
  }
  if (bad_dims("conv1_1_backward", n, H, W, NQA_PREC_F32, 0)) return NQA_E_ARG;
  return conv1_1_backward(gm_nhwc, w_oihw_dev, n, H, W, g_image_nchw, static_cast<hipStream_t>(stream));
}

}  // extern "C"
