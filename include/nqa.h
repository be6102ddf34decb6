/*
 * nqa.h -- C ABI of libnqa_hip.so: the DISTS / A-DISTS hot path of kobejean/nerf-qa
 * as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no FFI: its boundary for this path is the Python callable
 * surface of two nn.Modules.  Each entry point below names
 * the reference lines whose arithmetic it replaces; nerf_qa_amd/ (the Python
 * shell that mirrors those modules) is the only intended caller and binds these
 * symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; every pointer marked "dev" is a device pointer owned by
 *     the caller (e.g. a torch tensor's data_ptr()); "host" pointers are host memory;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Kernels
 *     are enqueued on it and the call returns without synchronising;
 *   - return value: 0 on success, a negative NQA_E_* code on failure, with a
 *     message available from nqa_last_error() (thread-local);
 *   - the only mutable state is THREAD-LOCAL (the error string, the timing ring of nqa_timing_*
 *     and the tuning choice of nqa_set_conv_variant) plus idempotent per-device caches (kernel
 *     attributes, CU count), so calls are re-entrant per (thread, stream): one thread's tuning
 *     or timing never changes or observes another thread's launches.
 *
 * Layouts
 *   - images enter as the reference's tensors: float32 NCHW, values in [0,1];
 *   - activations inside the pyramid are NHWC in the element type of the chosen
 *     precision (`prec`): NQA_PREC_F32 / NQA_PREC_F32S float, NQA_PREC_BF16 bfloat16,
 *     NQA_PREC_F16 IEEE half.  All accumulation and all statistics are float32/float64;
 *   - in NQA_PREC_F32S the maps BETWEEN conv layers (the outputs of nqa_conv1_1, of the
 *     non-tapped conv layers and of nqa_l2pool) are "split16": per pixel and per group of 16
 *     channels 64 bytes [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15], hi = half(v), lo = half(v - hi),
 *     4 bytes per element like float; the tapped maps (conv layers 1, 3, 6, 9, 12 = relu1_2 ..
 *     relu5_3) are plain float.  nqa_split16_encode / _decode convert;
 *   - VGG weights are handed over once as a packed blob (nqa_pack_vgg_weights).
 */
#ifndef NQA_H
#define NQA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NQA_VERSION 1

/* NQA_PREC_F32S: float32 activations like NQA_PREC_F32, but conv layers 2..13 multiply on the f16
 * matrix cores with both operands split into (hi, lo) half pairs (3 MFMAs per product block,
 * ~2^-21 relative error per product) -- near-f32 results at a fraction of the f32-MFMA cost.
 *
 * NQA_PREC_F32M ("mixed", DISTS pyramid entry points only: nqa_pack_vgg_weights, nqa_workspace_bytes,
 * nqa_vgg_pyramid, nqa_dists_forward): stages 1..3 (conv layers 0..6) keep f16 NHWC activations and multiply them
 * with weights held as f16 (hi, lo) pairs -- TWO MFMAs per product, the weights' 11-bit rounding removed -- and
 * stages 4..5 (layers 7..12) run as NQA_PREC_F32S; the L2-pool after stage 3 turns the f16 tap into split16
 * records.  Taps 1..3 are half, taps 4..5 float.  What is left of the 16-bit error is the activation rounding of the
 * first seven layers, whose contribution to a DISTS score is the smallest of all (tools/cpu_prec_layers.py).
 *
 * NQA_PREC_F32M2: the same with only stages 1..2 (conv layers 0..3) on two-term weights and stages 3..5 as
 * NQA_PREC_F32S (taps 1..2 half, 3..5 float): 2.6 times less of that residual error for ~10 % of the speed.
 * NQA_PREC_F32M4: stages 1..4 (layers 0..9) on two-term weights, stage 5 as NQA_PREC_F32S.  NQA_PREC_F16W: ALL five stages
 * -- f16 activations x two-term weights throughout, no float stage, every tap half: f16 without its weight rounding. */
enum { NQA_PREC_F32 = 0, NQA_PREC_BF16 = 1, NQA_PREC_F16 = 2, NQA_PREC_F32S = 3, NQA_PREC_F32M = 4, NQA_PREC_F32M2 = 5,
       NQA_PREC_F32M4 = 6, NQA_PREC_F16W = 7 };
/* pyramid stages (1-based count) that run with f16 activations + two-term weights in a mixed mode, 0 otherwise */
#define NQA_MIXED_STAGES(prec) \
  ((prec) == NQA_PREC_F32M ? 3 : (prec) == NQA_PREC_F32M2 ? 2 : (prec) == NQA_PREC_F32M4 ? 4 : (prec) == NQA_PREC_F16W ? 5 : 0)

enum {
  NQA_OK = 0,
  NQA_E_ARG = -1,       /* bad argument (null pointer, non-positive size, unknown prec) */
  NQA_E_SHAPE = -2,     /* shape the kernels do not support */
  NQA_E_WORKSPACE = -3, /* workspace too small */
  NQA_E_LAUNCH = -4     /* HIP reported an error when enqueuing */
};

#define NQA_NUM_CONVS 13
#define NQA_NUM_TAPS 6
#define NQA_TOTAL_CHNS 1475 /* 3+64+128+256+512+512, DISTS_pt.py:57 */

int nqa_version(void);
const char *nqa_last_error(void);

/* ---- VGG-16 weights -------------------------------------------------------- */

/* Bytes of the packed weight blob for `prec`. */
size_t nqa_packed_weights_bytes(int prec);

/* Pack the 13 conv3x3 layers (torchvision features[0,2,5,7,10,12,14,17,19,21,24,26,28],
 * sliced into stages at DISTS_pt.py:36-49) from float32 OIHW host arrays into the
 * kernel-native blob (host memory, nqa_packed_weights_bytes(prec) bytes).  The caller
 * then copies the blob to the device once. */
int nqa_pack_vgg_weights(const float *const w_host[NQA_NUM_CONVS], const float *const b_host[NQA_NUM_CONVS],
                         int prec, void *packed_host);

/* ---- single operators (used by forward_once and by the parity tests) ------- */

/* conv1_1 with the input normalisation folded in front: h=(x-mean)/std (DISTS_pt.py:92),
 * zero padding applied to h, conv3x3(3->64)+bias+ReLU (features[0,1]).  x: dev float32
 * NCHW (n,3,H,W); out: dev NHWC (n,H,W,64) in prec's element type (split16 in NQA_PREC_F32S). */
int nqa_conv1_1(const float *x_nchw, int n, int H, int W, const void *packed_w, int prec, void *out_nhwc,
                void *stream);

/* Stage 1 in one kernel (16-bit modes and NQA_PREC_F32S): conv1_1 as above followed by conv1_2+ReLU
 * (features[0..3], DISTS_pt.py:36-37), the 64-channel intermediate staying in LDS.
 * x: dev float32 NCHW (n,3,H,W); out: dev NHWC (n,H,W,64) = relu1_2 in prec's element type (float in NQA_PREC_F32S, whose
 * products are three-term (hi, lo) splits in both convolutions).  NQA_PREC_F32 has no fused form. */
int nqa_conv1_fused(const float *x_nchw, int n, int H, int W, const void *packed_w, int prec, void *out_nhwc,
                    void *stream);

/* conv3x3 stride 1 pad 1 + bias + ReLU for VGG layer `layer` (1..12), NHWC in/out
 * (torchvision Conv2d+ReLU pairs, DISTS_pt.py:36-49).  NQA_PREC_F32S: split16 in; float out for
 * the tapped layers (1, 3, 6, 9, 12), split16 out otherwise. */
int nqa_conv3x3_relu(const void *in_nhwc, int n, int H, int W, int layer, const void *packed_w, int prec,
                     void *out_nhwc, void *stream);

/* L2pooling.forward, DISTS_pt.py:22-25 (= Downsample, ADISTS.py:28-31):
 * sqrt(depthwise 3x3 Hanning, stride 2, pad 1, of x^2, + 1e-12).  NHWC (n,H,W,C) ->
 * (n,ceil(H/2),ceil(W/2),C).  NQA_PREC_F32S: float in (a tapped map), split16 out. */
int nqa_l2pool(const void *in_nhwc, int n, int H, int W, int C, int prec, void *out_nhwc, void *stream);

/* NHWC (prec element type) -> float32 NCHW, so forward_once can return the
 * reference's tensor format (DISTS_pt.py:103). */
int nqa_nhwc_to_nchw_f32(const void *in_nhwc, int n, int H, int W, int C, int prec, float *out_nchw, void *stream);

/* float32 NHWC (pixels, C) <-> split16 (see Layouts), C a multiple of 16.  Used by the
 * single-operator parity tests in NQA_PREC_F32S; the fused paths never call them. */
int nqa_split16_encode(const float *in_nhwc, long pixels, int C, void *out, void *stream);
int nqa_split16_decode(const void *in, long pixels, int C, float *out_nhwc, void *stream);

/* ---- the fused paths --------------------------------------------------------- */

/* Workspace bytes needed by nqa_vgg_pyramid / nqa_dists_forward for `n_images`
 * images (2*B for a batch of B pairs) of H x W. */
size_t nqa_workspace_bytes(int n_images, int H, int W, int prec);

/* forward_once, DISTS_pt.py:91-103, for n images: runs the 13 convs and 4 L2-pools.
 * taps[k] (k=0..4, dev, may not be null) receives relu{1_2,2_2,3_3,4_3,5_3} as NHWC
 * in prec's element type with shape (n, Hk, Wk, Ck), Hk = ceil(H / 2^k). */
int nqa_vgg_pyramid(const float *x_nchw, int n, int H, int W, const void *packed_w, int prec, void *workspace,
                    size_t workspace_bytes, void *const taps[5], void *stream);

/* DISTS.forward up to the per-channel similarities, DISTS_pt.py:105-141:
 * both pyramids (x and y, B images each, float32 NCHW (B,3,H,W)), then for every
 * (b, stage, channel) S1 = (2 mx my + 1e-6)/(mx^2 + my^2 + 1e-6) and
 * S2 = (2 cov + 1e-6)/(vx + vy + 1e-6).  s1, s2: dev float32 (B, 1475). */
int nqa_dists_forward(const float *x_nchw, const float *y_nchw, int B, int H, int W, const void *packed_w, int prec,
                      void *workspace, size_t workspace_bytes, float *s1, float *s2, void *stream);

/* The statistics alone on caller-provided float32 NCHW feature lists
 * (forward_from_feats, DISTS_pt.py:181-202).  fx[k], fy[k]: dev (B, C[k], Hk[k], Wk[k]).
 * scratch: dev, nqa_stats_scratch_bytes(B, total pixels...) bytes. */
size_t nqa_stats_scratch_bytes(int B, const int C[NQA_NUM_TAPS], const int Hk[NQA_NUM_TAPS], const int Wk[NQA_NUM_TAPS]);
int nqa_dists_stats_nchw(const float *const fx[NQA_NUM_TAPS], const float *const fy[NQA_NUM_TAPS], int B,
                         const int C[NQA_NUM_TAPS], const int Hk[NQA_NUM_TAPS], const int Wk[NQA_NUM_TAPS],
                         void *scratch, size_t scratch_bytes, float *s1, float *s2, void *stream);

/* alpha/beta weighted sum -> score, DISTS_pt.py:127-129,135,142,144:
 * w = sum(alpha)+sum(beta); score_b = 1 - sum_c alpha_c/w S1_bc - sum_c beta_c/w S2_bc.
 * alpha, beta: dev float32 (1475).  score: dev float32 (B). */
int nqa_dists_score(const float *s1, const float *s2, const float *alpha, const float *beta, int B, float *score,
                    void *stream);

/* ---- A-DISTS (ADISTS.forward, ADISTS.py:137-197, as_map=False) ------------------ */

size_t nqa_adists_workspace_bytes(int B, int H, int W, int prec);

/* Both pyramids, texture-probability maps from x (compute_prob, ADISTS.py:71-100),
 * entropy channel weights from x (ADISTS.py:127-135,150-161), Gaussian-windowed (21x21,
 * sigma 7, valid) or global T/S statistics per stage (ADISTS.py:165-183) and the
 * weighted combine (ADISTS.py:185-191).  d: dev float32 (B) receives D (the caller
 * returns 1-D or 1-mean(D), ADISTS.py:194-197). */
int nqa_adists_forward(const float *x_nchw, const float *y_nchw, int B, int H, int W, const void *packed_w, int prec,
                       void *workspace, size_t workspace_bytes, float *d, void *stream);

/* The same pass with as_map=True (ADISTS.py:163,188-189,193): additionally map[b] (dev float32
 * (B,H,W)) = 1 - sum over stages of the stage's distortion map resized bilinearly
 * (align_corners=False) to H x W.  The reference's return value broadcasts this to (B,B,H,W)
 * with out[i][j] = map[i] for every j (its (B,H,W) + (B,1,H,W) addition); the Python shell
 * reproduces that shape. */
int nqa_adists_forward_map(const float *x_nchw, const float *y_nchw, int B, int H, int W, const void *packed_w,
                           int prec, void *workspace, size_t workspace_bytes, float *d, float *map, void *stream);

/* ---- input preparation on the device (decoded uint8 frame -> metric input) ------------ */

/* transforms.ToTensor / `torch.from_numpy(frame).permute(2,0,1).float() / 255.0` (prep.py:89,
 * data.py:80): in dev uint8 (n,H,W,3) -> out dev float32 (n,3,H,W).  pil_roundtrip != 0 also
 * applies prep.py:90-91's ToPILImage -> ToTensor round trip (mul(255).byte() truncates). */
int nqa_u8hwc_to_f32nchw(const uint8_t *in, int n, int H, int W, int pil_roundtrip, float *out, void *stream);

/* F.interpolate(x, size=(Hout,Wout), mode='bilinear', align_corners=False) (prep.py:93-95,
 * data.py:81-82, test2_prep.py:437) on `planes` = n*C float32 planes of Hin x Win. */
int nqa_resize_bilinear_f32(const float *in, int planes, int Hin, int Win, int Hout, int Wout, float *out,
                            void *stream);

/* The two steps above fused (uint8 (n,Hin,Win,3) -> float32 (n,3,Hout,Wout)), bit-identical to
 * running them back to back; reads only the taps it needs. */
int nqa_u8_resize_bilinear_f32(const uint8_t *in, int n, int Hin, int Win, int Hout, int Wout, float *out,
                               void *stream);

/* transforms.functional.resize on a PIL image (DISTS_pt.py:213-215, test2_prep.py:112,225) =
 * PIL Image.resize((Wout,Hout), BILINEAR): Pillow's antialiased two-pass 8-bit resampler,
 * bit-exact.  in dev uint8 (n,Hin,Win,3) -> out dev uint8 (n,Hout,Wout,3). */
size_t nqa_resize_pil_workspace_bytes(int n, int Hin, int Win, int Hout, int Wout);
int nqa_resize_pil_bilinear_u8(const uint8_t *in, int n, int Hin, int Win, int Hout, int Wout, void *workspace,
                               size_t workspace_bytes, uint8_t *out, void *stream);

/* ---- tuning hook ------------------------------------------------------------------ */

/* The stage-closing conv of the DISTS path fused with what consumes its tap (nqa_conv_pool.hip; replaces
 * nerf_qa/DISTS_pytorch/DISTS_pt.py:94 `stage2` conv2_2 + ReLU, :22-25 the L2pooling in front of stage 3, and the
 * sums behind :130-142 for tap relu2_2): the tap is never written.  `in`: dev NHWC batch of 2B images of `layer`'s input
 * (x images [0,B), y images [B,2B)), 16-bit activations of `prec`'s stage; pooled: dev (2B, ceil(H/2), ceil(W/2), Cout)
 * in the next stage's input format; sums: dev double (B, Cout, 5) = {sum x, sum y, sum x^2, sum y^2, sum xy} over the
 * H*W pixels of the (rounded) tap, per pair and channel.  Only layer 3 (conv2_2) in NQA_PREC_F16 and the mixed modes
 * whose stage 3 stays 16-bit has a fused form so far: anything else returns NQA_E_SHAPE.  nqa_dists_forward takes this
 * path by itself (nqa_set_conv_variant + 64 turns it off); these two entry points exist for tests and tools. */
size_t nqa_conv_pool_workspace_bytes(int B, int H, int W, int layer);

/* Which tapped maps nqa_dists_forward(B, H, W, prec) closes inside their conv kernel (pool + statistics fused, the
 * full-resolution map never written): fused[k] = 1 for tap k (1..5; fused[0] is the image, always 0).  Depends on the
 * shape, the mode and the calling thread's nqa_set_conv_variant bits 64 / 128.  bench.py prices its HBM roofline
 * (the pool_stats passes that remain) with it. */
int nqa_dists_fused_taps(int B, int H, int W, int prec, int fused[6]);
/* The same for the WHOLE of stage 1 (nqa_conv1_pool.hip; DISTS_pt.py:92-94 normalisation + conv1_1 + conv1_2, the L2pooling
 * in front of stage 2 and tap relu1_2's sums) from the raw frames: x, y dev float32 NCHW (B,3,H,W); pooled: dev NHWC f16
 * (2B, ceil(H/2), ceil(W/2), 64), x images first; sums: dev double (B, 64, 5); workspace nqa_conv_pool_workspace_bytes(B, H,
 * W, 1).  NQA_PREC_F16 only so far (NQA_E_SHAPE otherwise).  The sums and the pool take relu1_2 rounded to f16, the values
 * the unfused kernels store and read back. */
int nqa_conv1_pool_stats(const float *x, const float *y, int B, int H, int W, const void *packed, int prec, void *pooled,
                         double *sums, void *ws, size_t ws_bytes, void *stream);
int nqa_conv_pool_stats(const void *in, int B, int H, int W, int layer, const void *packed, int prec, void *pooled,
                        double *sums, void *ws, size_t ws_bytes, void *stream);

/* Block-tile choice of the implicit-GEMM conv: 0 = 4-wave tiles (128 ch x 128 px) on every
 * layer, 1 (default) = + 8-wave 256 ch x 256 px tiles on layers with >= 256 output channels, 2 =
 * + 8-wave 128 ch x 512 px tiles wherever the map is large enough (measured equal to 1).
 * Adding 4 selects the tile form of the fused stage-1 kernel; adding 16 selects the round-1 forms of stage 1 (the
 * persistent two-phase kernel) and of conv2_1 (the implicit GEMM) instead of the register-resident-weights kernels, and
 * in NQA_PREC_F32S the round-2 pair of stage-1 kernels (VALU conv1_1 + implicit-GEMM conv1_2) instead of the fused one,
 * adding 32 the implicit GEMM for conv2_2 / conv3_1; adding 64 runs the DISTS path's tap 2 UNFUSED (conv2_2, then the
 * pool + statistics pass over the tap it wrote) instead of conv + L2-pool + statistics in one kernel (nqa_conv_pool.hip);
 * adding 128 does the same for stage 1 (conv1_regw_kernel + pool_stats_kernel instead of nqa_conv1_pool.hip's kernel);
 * adding 8 selects the first form of the A-DISTS window pass (every wave loads its own taps instead of sharing them
 * through LDS).  Results agree in every variant to the rounding of a different summation order inside a layer (the
 * tile variants are bit-identical); this only exists so they can be timed against each other in one process.
 * The choice is thread-local (it applies to the calling thread's later calls only). */
int nqa_set_conv_variant(int variant);

/* ---- per-kernel timing (bench.py's roofline leg) -------------------------------- */

/* When enabled, every launch of the conv / pool / stats kernels is bracketed by a
 * pair of hipEvents recorded on the launch stream.  nqa_timing_collect synchronises
 * those events and returns, per kernel class, the number of launches and the summed
 * device time in milliseconds, then clears the ring.  Thread-local: only the enabling thread's
 * launches are bracketed, and it collects only its own. */
enum { NQA_K_CONV1 = 0, NQA_K_CONV = 1, NQA_K_POOL = 2, NQA_K_STATS = 3, NQA_K_ADISTS = 4, NQA_K_PREP = 5, NQA_K_SEAM = 6,
       NQA_K_COUNT = 7 };  /* NQA_K_CONV holds the fused conv + pool + statistics kernels too; NQA_K_SEAM their seam pass */
int nqa_timing_enable(int on);
int nqa_timing_collect(int launches[NQA_K_COUNT], double ms[NQA_K_COUNT]);

/* ---- backward pass of the DISTS pyramid: DISTS.forward(x, y, require_grad=True), nerf_qa/DISTS_pytorch/DISTS_pt.py:105-108
 * (the reference runs forward_once WITH autograd there).  nerf_qa_amd/autograd.py drives these per layer; float
 * precision throughout (three-term split products).  Not on the scoring hot path.
 *   nqa_pack_conv_split      one 3x3 layer (float32 OIHW host array; cout % 64 == 0, cin % 16 == 0) in the NQA_PREC_F32S row
 *                            format, zero bias -- e.g. a layer's flipped, transposed weights for its data gradient;
 *   nqa_conv3x3_split        that layer: split16 NHWC in, FLOAT NHWC out, ReLU optional;
 *   nqa_relu_mask_split16    g * (act > 0) as split16 records (act: a float tapped map, or split16);
 *   nqa_l2pool_backward      g_tap += d(L2-pool)/d(tap) applied to g_pooled (DISTS_pt.py:22-25), pooled = the forward's split16 map;
 *   nqa_conv1_1_backward     g * (relu1_1 > 0) (float NHWC, 64 channels) -> gradient of the RAW image, float NCHW (n,3,H,W),
 *                            the (x - mean) / std of DISTS_pt.py:92 included; w = conv1_1's float32 OIHW weights on the device. */
size_t nqa_packed_conv_split_bytes(int cout, int cin);
int nqa_pack_conv_split(const float *w_oihw, int cout, int cin, void *packed_host);
int nqa_conv3x3_split(const void *in_split16, int n, int H, int W, int cin, int cout, const void *packed_conv, int relu,
                      float *out_nhwc, void *stream);
int nqa_relu_mask_split16(const float *g_nhwc, const void *act_nhwc, int act_is_split16, long pixels, int C,
                          void *out_split16, void *stream);
int nqa_l2pool_backward(const float *tap_nhwc, const void *pooled_split16, const float *g_pooled_nhwc, int n, int H, int W,
                        int C, float *g_tap_nhwc, void *stream);
int nqa_conv1_1_backward(const float *gm_nhwc, const float *w_oihw_dev, int n, int H, int W, float *g_image_nchw,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NQA_H */
