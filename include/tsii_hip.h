/*
 * tsii_hip.h -- C ABI of libtsii_hip.so, the MI355X (gfx950) kernels behind the
 * partial-convolution inpainting hot path of yu45020/Text_Segmentation_Image_Inpainting.
 *
 * The reference has no FFI/plugin seam (pure Python nn.Modules calling aten ops), so this
 * ABI is the seam a replacement .so provides underneath the reference's nn.Module surface;
 * each entry point names the reference lines whose aten ops it replaces (paths relative to
 * the reference repo root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - fp32 everywhere.  Activations are NHWC-contiguous ([N,H,W,C]; a 1x1 conv sees them as
 *    a row-major [M=N*H*W, C] matrix).  Mask "planes" are [N,H,W] fp32 holding exactly 0/1
 *    (or small integer counts); a channel-constant mask [N,C,H,W] is represented by its plane.
 *  - The caller owns every buffer (outputs, workspaces); nothing is allocated, freed or
 *    retained by the library.  All calls are asynchronous on `stream` (a hipStream_t).
 *  - Return value: 0 = enqueued; negative = invalid argument / unsupported shape / HIP
 *    launch error, message via tsii_last_error() (thread-local).  Never throws.
 *  - "row scale" (r0, split, r1): element (row m, channel k) of the operand is multiplied by
 *    r0[m] if k < split else r1[m]; r0 == NULL disables it; r1 == NULL means 1.0 for k >= split.
 *    This is how x*mask (partial_convolution.py:51,123) is fused for masks made of one or
 *    two channel-constant planes (decoder concat of up-sampled + skip masks,
 *    image_inpainting.py:83-84).
 */
#ifndef TSII_HIP_H
#define TSII_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSII_ABI_VERSION 5

/* activation kinds for the BN/activation kernels */
#define TSII_ACT_NONE 0
#define TSII_ACT_RELU 1
#define TSII_ACT_LEAKY 2 /* slope argument */
#define TSII_ACT_RELU6 3
#define TSII_ACT_SIGMOID 4

int tsii_version(void);
const char* tsii_last_error(void);

/* Arithmetic of the point-wise / implicit-GEMM matrix products.  The switch is THREAD-LOCAL: it applies to the entry points
 * the calling thread calls afterwards and nothing another thread does can change it (the only other state the library keeps is
 * the thread-local error string and a read-only device-properties cache).  A host that runs forward and backward on
 * different threads (PyTorch autograd does) sets it on each of them -- the package's `_lib.call` does so before every call.
 *   6 (default) split-bf16: each fp32 operand is split exactly into 3 bf16 pieces while it is staged, the 6 partial
 *               products of weight >= 2^-16 go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- dropped terms
 *               <= 2^-23 |a*b|, i.e. fp32-class results at 2.7x the matrix-core rate of the f32-input MFMA;
 *   8           the same with the two 2^-24 cross terms as well (only the 2^-32 term is dropped): 2x the f32 MFMA rate;
 *   3           2 pieces / 3 partial products (error <= 2^-15 |a*b|): inference-grade, opt-in;
 *   1           operands rounded to bf16, one product (fp32 accumulation, fp32 storage everywhere else): the "mixed bf16"
 *               arithmetic of BASELINE config 5, tolerance 1e-2 class, opt-in;
 *   0           v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chain).
 * The environment variable TSII_GEMM_PRODUCTS sets every thread's initial value; -1 puts the calling thread back to it.
 * inputs/outputs are fp32 in every mode.
 * Range caveat of the split modes: an operand that is inf, or finite but beyond the largest bf16 (3.39e38), splits into
 * (inf, NaN, NaN) -- where the f32 MFMA mode gives +-inf for that row, these give NaN; both rows are lost either way. */
int tsii_set_gemm_products(int products);
int tsii_get_gemm_products(void);

/* ---- K1: mask bookkeeping (partial_convolution.py:57-66,74-77,129-135) --------------- */

/* plane[n,h,w] = sum_c mask[n,c,h,w]; mask given with element strides (any layout).
 * With c == 1 this extracts channel 0 (the `mask[:, :1]` of :59 / :104). */
int tsii_mask_channel_sum(const float* mask, int n, int h, int w, int c,
                          int64_t sn, int64_t sh, int64_t sw, int64_t sc,
                          float* plane, void* stream);

/* S = a0*p0 + a1*p1 (p1 may be NULL); cnt = box_{kh x kw, stride, pad, dilation}(S) with zero
 * padding (padding counts as hole); hole = (cnt == 0).
 *   fill_holes != 0 (PartialConv :60-66,74-75): denom = hole ? 1 : cnt*post_scale,
 *                                               new_mask = hole ? 0 : 1, inv = hole ? 0 : 1/denom
 *   fill_holes == 0 (PartialConvNoHoles :130-135): denom = cnt*post_scale, new_mask = 1, inv = 1/denom
 * post_scale = Cin for same_holes (:61), 1 otherwise.  All values are small integers: bit-exact.
 * Any of denom / new_mask / inv may be NULL. */
int tsii_mask_update(const float* p0, float a0, const float* p1, float a1,
                     int n, int h, int w,
                     int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                     int ho, int wo, float post_scale, int fill_holes,
                     float* denom, float* new_mask, float* inv, void* stream);

/* nearest x2 up-sampling of a plane (DoubleUpSample on the mask, partial_convolution.py:231) */
int tsii_plane_upsample2x(const float* in, int n, int h, int w, float* out, void* stream);

/* out = x * mask, both NHWC with identical shape (general per-channel masks, :51) */
int tsii_mul_mask(const float* x, const float* mask, int64_t numel, float* out, void* stream);
/* dx = dy * mask */
/* (same entry point: multiplication is its own adjoint) */

/* ---- K3: point-wise (1x1) convolution as an MFMA GEMM ---- ------------------------------
 * PartialConv1x1 (:101-105), PartialConvNoHoles k=1 (:121-137), PartialConv k=1.
 *   y[m,n] = keep[m] ? (sum_k x[m,k]*rs(m,k)*w[n,k]) / denom[m] + bias[n] : 0
 * denom/keep/bias/r0 may be NULL (plain conv). */
/* ws: weight workspace of tsii_pw_ws_bytes(n, k) bytes -- in the split-bf16 arithmetic modes (tsii_set_gemm_products)
 * the weights are split into bf16 planes there once per call (otherwise every block splits its weight tile again while
 * staging it); NULL is allowed. */
size_t tsii_pw_ws_bytes(int n, int k);
int tsii_pw_fwd(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                const float* r0, int split, const float* r1,
                const float* denom, const float* keep, float* y, void* ws, size_t ws_bytes, void* stream);
/* dx[m,k] = rs(m,k) * sum_n dy[m,n]*inv[m]*w[n,k];  wt_ws: tsii_pw_ws_bytes(n, k) bytes of scratch (required) */
int tsii_pw_bwd_dx(const float* dy, int64_t m, int n, const float* w, int k, const float* inv,
                   const float* r0, int split, const float* r1, float* dx, float* wt_ws, void* stream);
/* dw[n,k] = sum_m dy[m,n]*inv[m] * x[m,k]*rs(m,k);  dbias[n] = sum_m dy[m,n]*keep[m] (dbias may be
 * NULL; keep NULL = all rows: the bias is added after the division, so its gradient is not scaled) */
size_t tsii_pw_bwd_dw_ws_bytes(int64_t m, int n, int k);
int tsii_pw_bwd_dw(const float* dy, const float* x, int64_t m, int n, int k, const float* inv, const float* keep,
                   const float* r0, int split, const float* r1, float* dw, float* dbias,
                   void* ws, size_t ws_bytes, void* stream);

/* ---- K2: depth-wise partial convolution (PartialConv groups=C, MobileNetV2.py:174-176) --
 *   y[n,ho,wo,c] = keep ? (sum_taps w[c,t]*x[n,hi,wi,c]*rmask[n,hi,wi]) / denom[n,ho,wo] + b[c] : 0
 * w is the reference layout [C,1,kh,kw]; ws: c*kh*kw floats of scratch. */
int tsii_dw_fwd(const float* x, const float* rmask, const float* w, const float* bias,
                const float* denom, const float* keep,
                int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                int ho, int wo, float* y, float* ws, void* stream);
int tsii_dw_bwd_dx(const float* dy, const float* inv, const float* w, const float* rmask,
                   int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                   int ho, int wo, float* dx, float* ws, void* stream);
size_t tsii_dw_bwd_dw_ws_bytes(int n, int ho, int wo, int c, int kh, int kw);
int tsii_dw_bwd_dw(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                   int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                   int ho, int wo, float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- K4: dense k x k partial convolution, groups == 1 (PartialConv.forward :49-80) ------
 * x*mask is either the two-plane row scale (r0/split/r1 at input resolution) or a full
 * per-channel mask `mfull` (same NHWC shape as x; ImageFill stem, image_inpainting.py:23).
 * w is the reference layout [Cout,Cin,kh,kw]. */
size_t tsii_dense_ws_bytes(int cin, int cout, int kh, int kw);
int tsii_dense_fwd(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                   const float* w, const float* bias, const float* denom, const float* keep,
                   int n, int h, int wd, int cin, int cout,
                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                   float* y, void* ws, size_t ws_bytes, void* stream);
int tsii_dense_bwd_dx(const float* dy, const float* inv, const float* w,
                      const float* mfull, const float* r0, int split, const float* r1,
                      int n, int h, int wd, int cin, int cout,
                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                      float* dx, void* ws, size_t ws_bytes, void* stream);
size_t tsii_dense_bwd_dw_ws_bytes(int n, int ho, int wo, int cin, int cout, int kh, int kw);
int tsii_dense_bwd_dw(const float* dy, const float* inv, const float* keep, const float* x,
                      const float* mfull, const float* r0, int split, const float* r1,
                      int n, int h, int wd, int cin, int cout,
                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                      float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- K4c: the decoder's last level without its concatenated tensor: 3x3 / stride 1 / pad 1 PartialConv with <= 4 output
 * channels over cat(nearest-x2(low [n,h/2,wd/2,c1]), skip [n,h,wd,c2])  (DoubleUpSample + torch.cat + the 35 -> 3 output layer,
 * models/partial_convolution.py:224-231, models/image_inpainting.py:82-86).  r0 / r1: mask planes [n,h,wd] of the two parts
 * (NULL = ones; the x*mask split is the concat boundary c1), denom / keep / inv as for tsii_dense_*.  Equal to
 * tsii_upcat_fwd + tsii_dense_fwd (and tsii_dense_bwd_dx + tsii_upcat_bwd, tsii_dense_bwd_dw) to rounding.
 * tsii_head_cat_ok() != 0: this geometry has the fused kernels (h, wd even, c1 % 4 == 0, 20 < c1 + c2 <= 68, cout <= 4).
 * ws: tsii_dense_ws_bytes(c1 + c2, cout, 3, 3) (fwd, bwd_dx) / tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3) (bwd_dw).
 * dskip may be NULL (the skip is a data tensor). */
int tsii_head_cat_ok(int n, int h, int wd, int c1, int c2, int cout);
int tsii_head_cat_fwd(const float* low, const float* skip, int c1, int c2, const float* r0, const float* r1,
                      const float* w, const float* bias, const float* denom, const float* keep,
                      int n, int h, int wd, int cout, float* y, void* ws, size_t ws_bytes, void* stream);
int tsii_head_cat_bwd_dx(const float* dy, const float* inv, const float* w, int c1, int c2, const float* r0, const float* r1,
                         int n, int h, int wd, int cout, float* dlow, float* dskip, void* ws, size_t ws_bytes, void* stream);
int tsii_head_cat_bwd_dw(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                         int c1, int c2, const float* r0, const float* r1, int n, int h, int wd, int cout,
                         float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* K4d: tsii_head_cat_bwd_dw on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32), with the low part's mask given where
 * it lives: r0_low [n, h/2, wd/2] is the mask plane of `low` itself (NULL = ones) -- the up-sampled half is then constant over
 * each 2x2 block and its share of the product runs over LOW-resolution pixels against 2x2 box sums of dy*inv.
 * tsii_head_cat_low_ok() != 0: h % 16 == 0, wd % 64 == 0 (whole 16 x 64 tiles), c1 in {32, 64}, 1 <= c2 <= 16, cout <= 3.
 * Same workspace as tsii_head_cat_bwd_dw. */
int tsii_head_cat_low_ok(int n, int h, int wd, int c1, int c2, int cout);
/* ... the weight gradient AND the gradient of `low` in one pass (the box sums the former holds in LDS are the left operand of the
 * latter: dlow[j][ci] = r0_low[j] * sum_m S[j][m] W[m][ci]); tsii_head_cat_bwd_low_ok(): tsii_head_cat_low_ok() and c1 == 32.
 * = tsii_head_cat_bwd_dw_low + the d low half of tsii_head_cat_bwd_dx (no d skip: the skip is a data tensor). */
int tsii_head_cat_bwd_low_ok(int n, int h, int wd, int c1, int c2, int cout);
int tsii_head_cat_bwd_low(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                          int c1, int c2, const float* r0_low, const float* r1, const float* w, int n, int h, int wd, int cout,
                          float* dwgt, float* dbias, float* dlow, void* ws, size_t ws_bytes, void* stream);
/* ... and the forward pass: the up-sampled half as Z[low pixel][(tap, cout)] = W_low x low on the matrix cores (kept in LDS per
 * tile), 9 entries of Z per output pixel and channel + the 3-channel skip half on the vector ALU.  tsii_head_cat_fwd_low_ok():
 * tsii_head_cat_low_ok() and c2 == 3.  No workspace. */
int tsii_head_cat_fwd_low_ok(int n, int h, int wd, int c1, int c2, int cout);
int tsii_head_cat_fwd_low(const float* low, const float* skip, int c1, int c2, const float* r0_low, const float* r1,
                          const float* w, const float* bias, const float* denom, const float* keep,
                          int n, int h, int wd, int cout, float* y, void* stream);
int tsii_head_cat_bwd_dw_low(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                             int c1, int c2, const float* r0_low, const float* r1, int n, int h, int wd, int cout,
                             float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- K4b: stems (odd k, stride 2, pad (k-1)/2, very few input channels; models/image_inpainting.py:23) as a stride-1
 * valid convolution over the space-to-depth image, so they run on the vector-gather implicit GEMM:
 *   tsii_stem_s2d:   x [n,h,w,c] * mask (mfull, or the r0/split/r1 planes, or none) zero-padded by `pad`
 *                    -> out [n,(h+2pad)/2,(w+2pad)/2,4c], channel order (row phase, column phase, c)
 *   tsii_stem_w_fwd: w [cout,cin,k,k] -> w2 [cout,4cin,ka,ka], ka = (k+1)/2 (reference layout of that conv)
 *   tsii_stem_w_bwd: dw2 [cout,4cin,ka,ka] -> dw [cout,cin,k,k]
 * then tsii_dense_fwd / tsii_dense_bwd_dw on (out, w2) with kernel ka, stride 1, padding 0. */
int tsii_stem_s2d(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                  int n, int h, int w, int c, int pad, float* out, void* stream);
int tsii_stem_w_fwd(const float* w, int cout, int cin, int k, float* w2, void* stream);
int tsii_stem_w_bwd(const float* dw2, int cout, int cin, int k, float* dw, void* stream);

/* ---- K6: BatchNorm2d (+activation, +residual)  (partial_convolution.py:193-197,
 *          residual add MobileNetV2.py:186-187, image_inpainting.py:216) ------------------
 * y is [M,C].  Training: batch mean / biased variance (and running-stat update with
 * momentum, unbiased variance, like nn.BatchNorm2d); eval: pass the running stats to apply. */
size_t tsii_bn_ws_bytes(int64_t m, int c);
int tsii_bn_stats(const float* y, int64_t m, int c, float* mean, float* var,
                  float* running_mean, float* running_var, float momentum,
                  void* ws, size_t ws_bytes, void* stream);
/* out = act(gamma*(y-mean)/sqrt(var+eps)+beta) (+ residual) */
int tsii_bn_act_fwd(const float* y, int64_t m, int c, const float* mean, const float* var,
                    const float* gamma, const float* beta, float eps, int act, float slope,
                    const float* residual, float* out, void* stream);
/* training != 0: full batch-stat backward; else eval backward.  dgamma/dbeta always written. */
int tsii_bn_act_bwd(const float* dout, const float* y, int64_t m, int c,
                    const float* mean, const float* var, const float* gamma, const float* beta,
                    float eps, int act, float slope, int training,
                    float* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* ---- K6b: BatchNorm folded into the neighbouring convolutions (training-time fusion of the reference's
 *           Sequential(conv, BatchNorm2d, act, conv ...) chains, e.g. models/MobileNetV2.py:168-179) -----
 * Producer side ("stat_part"): the conv kernel also writes, per block of output rows and per channel,
 *   (count, p, sum(y - p), sum((y - p)^2)) with the pivot p a value of that block (no cancellation for
 *   near-constant channels), as float [rows][4][c_out]; rows = tsii_pw_stat_rows(m) /
 *   tsii_dw_stat_rows(...).  tsii_bn_finalize() combines them (fp64, parallel-variance formula) into the
 *   batch mean / biased variance, updates the running statistics like nn.BatchNorm2d and emits
 *   scale = gamma/sqrt(var+eps), shift = beta - mean*scale -- no separate pass over y.
 * Consumer side ("in_scale/in_shift"): the next conv applies a = act(in_scale[c]*v + in_shift[c]) to every
 *   element it loads (before the x*mask multiply; zero padding pads a), so the normalised activation is never
 *   written to HBM.  in_scale == NULL: plain input; in_act is NONE, RELU, LEAKY (slope in [0,1]) or RELU6 (the
 *   load-time form is min(max(z, neg*z), hi); anything else is an error).  The *_bwd_dw_bn forms recompute a the same way;
 *   dX is unchanged (it is the gradient w.r.t. a) and feeds tsii_bn_act_bwd together with the raw y. */
int64_t tsii_pw_stat_rows(int64_t m);
int tsii_pw_fwd_bn(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                   const float* r0, int split, const float* r1, const float* denom, const float* keep,
                   const float* in_scale, const float* in_shift, int in_act, float in_slope,
                   float* stat_part, float* y, void* ws, size_t ws_bytes, void* stream);
int tsii_pw_bwd_dw_bn(const float* dy, const float* x, int64_t m, int n, int k, const float* inv, const float* keep,
                      const float* r0, int split, const float* r1,
                      const float* in_scale, const float* in_shift, int in_act, float in_slope,
                      float* dw, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* 0 when the geometry has no LDS-tiled kernel (then only the unfused entry points apply) */
int64_t tsii_dw_stat_rows(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw);
int tsii_dw_fwd_bn(const float* x, const float* rmask, const float* w, const float* bias,
                   const float* denom, const float* keep,
                   int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                   int ho, int wo, const float* in_scale, const float* in_shift, int in_act, float in_slope,
                   float* stat_part, float* y, float* ws, void* stream);
int tsii_dw_bwd_dw_bn(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                      int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                      int ho, int wo, const float* in_scale, const float* in_shift, int in_act, float in_slope,
                      float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* dense k x k forward with the statistics partials ([rows][4][cout]); rows == 0: this geometry is not on the
 * implicit-GEMM path (few-output-channel / generic kernels), use tsii_dense_fwd + tsii_bn_stats */
int64_t tsii_dense_stat_rows(int has_mfull, int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw,
                             int ph, int pw, int dh, int dw, int ho, int wo);
int tsii_dense_fwd_bn(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                      const float* w, const float* bias, const float* denom, const float* keep,
                      int n, int h, int wd, int cin, int cout,
                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                      float* stat_part, float* y, void* ws, size_t ws_bytes, void* stream);
size_t tsii_bn_finalize_ws_bytes(int64_t rows, int c);
/* scale/shift may be NULL (then gamma/beta may be too) */
int tsii_bn_finalize(const float* stat_part, int64_t rows, int c, int64_t m,
                     float* mean, float* var, float* running_mean, float* running_var, float momentum,
                     const float* gamma, const float* beta, float eps, float* scale, float* shift,
                     void* ws, size_t ws_bytes, void* stream);
/* ---- K6c: the two reductions of the BatchNorm backward taken by the kernel that PRODUCES its incoming gradient.
 * tsii_dw_bwd_dx_bn: depth-wise dX (3x3; stride 1 / dilation 1, or stride 2 / pad 1 / dilation 1: the marching-strip paths,
 *   rows = tsii_dw_bwd_stat_rows(n,h,wd,c,...) > 0) whose output dx is the gradient w.r.t. a = act(BN(bn_y)), bn_y being the raw [n,h,wd,c] tensor the depth-wise conv consumed
 *   through its load-time BatchNorm; it also writes bwd_part[rows][2][c] = per-strip (sum dz, sum dz*xhat),
 *   dz = dx*act'(z).  tsii_bn_act_bwd_pre is tsii_bn_act_bwd without its reduction pass (ws: tsii_bn_ws_bytes(m, c)). */
int64_t tsii_dw_bwd_stat_rows(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);
int tsii_dw_bwd_dx_bn(const float* dy, const float* inv, const float* w, const float* rmask,
                      int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                      int ho, int wo, const float* bn_y, const float* bn_mean, const float* bn_var,
                      const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                      float* dx, float* bwd_part, float* ws, void* stream);
/* K6d: tsii_dw_bwd_dx_bn that ALSO returns the layer's weight gradient dwgt[c][1][3][3] (no bias gradient: the layer has no bias) --
 * the dX pass holds dy * inv with its halo in LDS and forms the layer's input act(BN(bn_y)) * rmask at its own pixels for the K6c
 * sums, i.e. both operands of dW; tsii_dw_bwd_dw_bn's second pass over (dy, bn_y) is not run.  3x3; stride 1 at dilation 1 / 2 / 4 / 8 (padding 0..2d; not the row-phase kernel's geometries) or stride 2 / padding 1 / dilation 1:
 * ws_dw of tsii_dw_bwd_dxdw_ws_bytes(...) bytes, 0 = no such form for this geometry (call the two separate entry points).
 * Replaces the autograd of F.conv2d(groups=C) in models/partial_convolution.py:49-51 for dX and dW together. */
size_t tsii_dw_bwd_dxdw_ws_bytes(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);
int tsii_dw_bwd_dxdw_bn(const float* dy, const float* inv, const float* w, const float* rmask,
                        int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                        int ho, int wo, const float* bn_y, const float* bn_mean, const float* bn_var,
                        const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                        float* dx, float* bwd_part, float* dwgt, float* ws, void* ws_dw, size_t ws_dw_bytes, void* stream);
/* K6e: tsii_dw_bwd_dxdw_bn fed with the gradient w.r.t. the ACTIVATION that follows the layer: da2 = d loss / d act2(BN2(y2)), y2 = the
 * layer's raw output (bn2_y, same [n,ho,wo,c] layout), bn2_coef[6][c] = (mean, 1/std, gamma, beta, dbeta/m, dgamma/m) of BN2 as
 * tsii_bn_bwd_reduce leaves it -- BN2's backward is applied while the kernel stages its slab, so tsii_bn_act_bwd_pre's pass over
 * (da2, y2) -> dy2 and this kernel's read of dy2 become this kernel's reads of da2 and y2.  Stride 1 / dilation 1 and stride 2 / padding 1
 * (tsii_dw_bwd_dxdw_fold_ok() == 1); everything else as tsii_dw_bwd_dxdw_bn.
 * tsii_bn_bwd_reduce / tsii_bn_bwd_apply: the two halves of tsii_bn_act_bwd_pre (reduction of the K6c partial rows to dgamma, dbeta
 * and the table; the stand-alone apply pass over the table: dy = ((dout act'(z) - coef[4]) - xhat coef[5]) gamma / std).
 * Together they replace the autograd of nn.BatchNorm2d + activation in models/partial_convolution.py:176-180 (bn_act). */
int tsii_dw_bwd_dxdw_fold_ok(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);
int tsii_dw_bwd_dxdw_bn2(const float* da2, const float* bn2_y, const float* bn2_coef, int bn2_act, float bn2_slope,
                         const float* inv, const float* w, const float* rmask,
                         int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                         int ho, int wo, const float* bn_y, const float* bn_mean, const float* bn_var,
                         const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                         float* dx, float* bwd_part, float* dwgt, float* ws, void* ws_dw, size_t ws_dw_bytes, void* stream);
size_t tsii_bn_bwd_reduce_ws_bytes(int64_t rows, int c);
int tsii_bn_bwd_reduce(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int training,
                       const float* bwd_part, int64_t rows, int64_t m, int c, float* dgamma, float* dbeta, float* coef,
                       void* ws, size_t ws_bytes, void* stream);
int tsii_bn_bwd_apply(const float* dout, const float* y, int64_t m, int c, const float* coef, int act, float slope, float* dy,
                      void* stream);
/* same for the point-wise dX (N = k % 4 == 0): dx is the gradient w.r.t. a = act(BN(bn_y)), bn_y raw [m,k];
 * bwd_part[tsii_pw_stat_rows(m)][2][k] */
int tsii_pw_bwd_dx_bn(const float* dy, int64_t m, int n, const float* w, int k, const float* inv,
                      const float* r0, int split, const float* r1,
                      const float* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                      const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                      float* dx, float* bwd_part, float* wt_ws, void* stream);
int tsii_bn_act_bwd_pre(const float* dout, const float* y, int64_t m, int c,
                        const float* mean, const float* var, const float* gamma, const float* beta,
                        float eps, int act, float slope, int training, const float* bwd_part, int64_t rows,
                        float* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* tsii_bn_act_bwd_pre that ALSO leaves pooled[n, y/2, x/2, :] = sum over the 2x2 block of dy[n, y, x, :] * pool_scale[n, y, x]
 * (rows of dy = pixels of [.., up_h, up_w] images, both even; pool_scale NULL = 1): dy is the gradient of a tsii_pw_fwd_up
 * output, pooled the gradient of its up-sampled addend (= tsii_pool2x2_scaled(dy, pool_scale), without the second pass). */
int tsii_bn_act_bwd_pre_pool(const float* dout, const float* y, int64_t m, int c,
                             const float* mean, const float* var, const float* gamma, const float* beta,
                             float eps, int act, float slope, int training, const float* bwd_part, int64_t rows,
                             int up_h, int up_w, const float* pool_scale, float* dy, float* pooled,
                             float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* eval mode: (scale, shift) from the running statistics */
int tsii_bn_scale_shift(const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                        int c, float* scale, float* shift, void* stream);
/* activation only (PartialActivation :204-211): out = act(x); dx = dout*act'(x) */
int tsii_act_fwd(const float* x, int64_t numel, int act, float slope, float* out, void* stream);
int tsii_act_bwd(const float* dout, const float* x, int64_t numel, int act, float slope, float* dx, void* stream);

/* ---- K7: nearest x2 up-sampling + channel concat (DoubleUpSample + torch.cat,
 *          partial_convolution.py:229-231, image_inpainting.py:82-85) --------------------
 * out[n,y,x,0:c1] = low[n,y/2,x/2,:], out[n,y,x,c1:] = skip[n,y,x,:];  low is [n,h,w,c1].
 * c2 == 0 (skip NULL) is the plain DoubleUpSample of x. */
int tsii_upcat_fwd(const float* low, const float* skip, int n, int h, int w, int c1, int c2,
                   float* out, void* stream);
int tsii_upcat_bwd(const float* dout, int n, int h, int w, int c1, int c2,
                   float* dlow, float* dskip, void* stream);

/* ---- K7b (round 4): the same DoubleUpSample + torch.cat in front of a 1x1 partial convolution, never written --------------
 * A 1x1 convolution commutes with nearest up-sampling and distributes over the channel concatenation:
 *   conv1x1(cat(up2(low), skip) * mask) = up2(conv1x1_low(low * mask_low)) + conv1x1_skip(skip * mask_skip)
 * (models/partial_convolution.py:121-137 over image_inpainting.py:82-85), so the decoder's expand convolutions run their
 * low-resolution half at LOW resolution -- a quarter of its multiply-adds, and the concatenated tensor never exists:
 *   z = tsii_pw_fwd(low, ..)                       plain [m/4, n] product at low resolution
 *   y = tsii_pw_fwd_up(skip, .., up_add = z)       y[row] = keep ? (acc[row] + z[low_row(row)]) / denom + bias : 0
 * rows of y are the pixels of [.., up_h, up_w] images (up_h even, up_w % 4 == 0, m % (up_h*up_w) == 0, n % 4 == 0, m < 2^31);
 * stat_part (or NULL) as in tsii_pw_fwd_bn.  Backward: tsii_pw_bwd_dx / tsii_pw_bwd_dw on the skip half as usual, and
 * dz = tsii_pool2x2_scaled(dy, inv) -- dlow[n,y,x,:] = sum of the 2x2 block of dout[n,2y+dy,2x+dx,:] * scale[n,2y+dy,2x+dx]
 * (scale NULL = 1; dout is [n,2h,2w,c]) -- is the gradient of z. */
int tsii_pw_fwd_up(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                   const float* r0, int split, const float* r1, const float* denom, const float* keep,
                   const float* up_add, int up_h, int up_w, float* stat_part, float* y, void* ws, size_t ws_bytes,
                   void* stream);
int tsii_pool2x2_scaled(const float* dout, const float* scale, int n, int h, int w, int c, float* dlow, void* stream);

/* ---- K11 (first piece): mean-L1 loss (nn.L1Loss, loss.py:190; bench loss of SURVEY 8d) - */
size_t tsii_l1_ws_bytes(int64_t numel);
int tsii_l1_mean_fwd(const float* a, const float* b, int64_t numel, float* loss,
                     void* ws, size_t ws_bytes, void* stream);
/* da = sign(a-b) * (*gscale) / numel;  gscale is a device pointer to the upstream scalar grad */
int tsii_l1_mean_bwd(const float* a, const float* b, int64_t numel, const float* gscale,
                     float* da, void* stream);

/* ---- training-step helpers ---------------------------------------------------------- */
/* fused SGD step (nesterov momentum + weight decay; the optimiser the reference trained with,
 * checkpoints/ReadME.md:4): g += wd*p; buf = mom*buf + g; p -= lr*(g + mom*buf) */
int tsii_sgd_nesterov(float* p, const float* g, float* buf, int64_t numel,
                      float lr, float momentum, float weight_decay, void* stream);

/* ---- segmentation path (models/common.py, models/text_segmentation.py, loss.py) ------- */
/* out = act(a + b): residual adds `x + self.conv(x)` (models/MobileNetV2.py:146-147, models/Xception.py:44)
 * and `act_fn(rfb_pool + resi)` (models/common.py:156).  Backward: tsii_act_bwd(dout, out, ...) twice. */
int tsii_add_act_fwd(const float* a, const float* b, int64_t numel, int act, float slope, float* out, void* stream);
/* channel concat / slice on [M, C] NHWC rows (torch.cat dim=1, models/text_segmentation.py:68,75,80,111;
 * models/common.py:91,151):  to_dst != 0: big[m, coff + c] = small[m, c];  else small[m, c] = big[m, coff + c] */
int tsii_copy_channels(float* big, int64_t m, int cbig, int coff, float* small_, int csmall, int to_dst, void* stream);
/* bilinear up-sampling by an integer factor, align_corners=False (F.interpolate / nn.Upsample,
 * models/text_segmentation.py:54,76,109,113); x is [n,h,w,c], y is [n,h*s,w*s,c] */
int tsii_bilinear_up_fwd(const float* x, int n, int h, int w, int c, int scale, float* y, void* stream);
int tsii_bilinear_up_bwd(const float* dy, int n, int h, int w, int c, int scale, float* dx, void* stream);
/* global average pool (nn.AdaptiveAvgPool2d(1), models/common.py:19,35): gap[n,c] = mean_hw x[n,hw,c] */
size_t tsii_gap_ws_bytes(int n, int hw, int c);
int tsii_gap_fwd(const float* x, int n, int hw, int c, float* gap, void* ws, size_t ws_bytes, void* stream);
int tsii_gap_bwd(const float* dgap, int n, int hw, int c, float* dx, void* stream);
/* scSE combine (models/common.py:38-43): out = x*cse[n,c] + x*sse[n,hw];
 * backward: dx = g*(cse+sse), dcse[n,c] = sum_hw g*x, dsse[n,hw] = sum_c g*x */
int tsii_scse_fwd(const float* x, const float* cse, const float* sse, int n, int hw, int c, float* out, void* stream);
int tsii_scse_bwd(const float* g, const float* x, const float* cse, const float* sse, int n, int hw, int c,
                  float* dx, float* dcse, float* dsse, void* ws, size_t ws_bytes, void* stream);
/* BinaryFocalLoss (loss.py:58-75): mean( exp(gamma*logsigmoid(-x*(2t-1))) * w*BCEwithlogits(x,t) ),
 * w = words_w if t > 0 else background_w.  ws: tsii_l1_ws_bytes(numel). */
int tsii_bce_focal_fwd(const float* x, const float* t, int64_t numel, float gamma, float background_w, float words_w,
                       float* loss, void* ws, size_t ws_bytes, void* stream);
int tsii_bce_focal_bwd(const float* x, const float* t, int64_t numel, float gamma, float background_w, float words_w,
                       const float* gscale, float* dx, void* stream);

/* pixel shuffle by r on NHWC (the reference only mentions it in its README; semantics =
 * torch.nn.PixelShuffle on NCHW): x [n,h,w,c*r*r] -> y [n,h*r,w*r,c], y[n,h*r+i,w*r+j,c] = x[n,h,w,c*r*r+i*r+j].
 * inverse != 0 runs the adjoint / un-shuffle (y -> x). */
int tsii_pixel_shuffle(const float* src, int n, int h, int w, int c, int r, int inverse, float* dst, void* stream);

/* ---- InpaintingLoss pieces (loss.py:195-225,303-307) ------------------------------------ */
/* comp = mask*raw + (1-mask)*out (loss.py:196); backward: dout = dcomp*(1-mask) */
int tsii_compose_fwd(const float* raw, const float* mask, const float* out, int64_t numel, float* comp, void* stream);
int tsii_compose_bwd(const float* dcomp, const float* mask, int64_t numel, float* dout, void* stream);
/* loss = w_valid*mean|m*out - m*gt| + w_hole*mean|(1-m)*out - (1-m)*gt|  (loss.py:199-200,223); ws: tsii_l1_ws_bytes */
int tsii_masked_l1_fwd(const float* out, const float* gt, const float* mask, int64_t numel, float w_valid, float w_hole,
                       float* loss, void* ws, size_t ws_bytes, void* stream);
int tsii_masked_l1_bwd(const float* out, const float* gt, const float* mask, int64_t numel, float w_valid, float w_hole,
                       const float* gscale, float* dout, void* stream);
/* total_variation_loss (loss.py:303-307) on NHWC [n,h,w,c]: mean|x[:,:,:,:-1]-x[:,:,:,1:]| + mean|x[:,:,:-1,:]-x[:,:,1:,:]| */
int tsii_tv_fwd(const float* x, int n, int h, int w, int c, float* loss, void* ws, size_t ws_bytes, void* stream);
int tsii_tv_bwd(const float* x, int n, int h, int w, int c, const float* gscale, float* dx, void* stream);

/* ==== bf16 ACTIVATION STORAGE (BASELINE config 5: "mixed bf16 with fp32 mask renorm"; round 5) ==========================
 * The segmentation nets (models/text_segmentation.py:18-114 and their blocks: models/MobileNetV2.py:114-149,
 * models/Xception.py:13-114, models/common.py:53-156, models/BaseModels.py:91-127) with activations AND activation
 * gradients kept in HBM as bf16 NHWC; parameters, parameter gradients, BatchNorm statistics, every accumulation and every
 * reduction stay fp32.  `uint16_t*` = raw bf16 bits; every channel count is a multiple of 8 (one 16-byte vector; the host
 * pads the 3-channel stem to 4 and a 1-channel head to 8), every tensor base is 16-byte aligned.  A kernel reads bf16,
 * computes in fp32 and rounds once (RNE) when it stores; the BatchNorm partials a kernel emits describe the ROUNDED values.
 * These nets carry no mask planes, so the entry points have none (the partial-convolution family keeps fp32 storage: its
 * count division / hole logic is the "fp32 mask renorm" of the config and no shipped model combines the two).
 * Layouts of the partial rows are those of the fp32 entry points above (stat_part [rows][4][c], bwd_part [rows][2][c]), so
 * tsii_bn_finalize and the fp32 reductions are shared. */

/* ---- matrix products: 1x1 convolutions (bf16 x bf16 -> fp32 on v_mfma_f32_32x32x16_bf16) ---------------------------- */
int64_t tsii_bf16_stat_rows(int64_t m);                 /* partial rows of the NT kernels: one per 128 output rows */
size_t tsii_bf16_pw_ws_bytes(int n, int k);             /* bf16 image of the weights, written once per call */
/* y[m,n] = sum_k a(x[m,k]) w[n,k] + bias[n],  a = act(in_scale*x + in_shift) applied while loading (in_scale NULL: a = x);
 * stat_part (or NULL): BatchNorm partials of y [tsii_bf16_stat_rows(m)][4][n] */
int tsii_bf16_pw_fwd(const uint16_t* x, int64_t m, int k, const float* w, int n, const float* bias,
                     const float* in_scale, const float* in_shift, int in_act, float in_slope,
                     float* stat_part, uint16_t* y, void* ws, size_t ws_bytes, void* stream);
/* dx[m,k] = sum_n dy[m,n] w[n,k]; with bn_y (raw [m,k] input of the BatchNorm the conv consumed on load, K6c) also
 * bwd_part[tsii_bf16_stat_rows(m)][2][k] = (sum dz, sum dz*xhat), dz = dx*act'(z).  ws: tsii_bf16_pw_ws_bytes(n, k). */
int tsii_bf16_pw_bwd_dx(const uint16_t* dy, int64_t m, int n, const float* w, int k,
                        const uint16_t* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                        const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                        uint16_t* dx, float* bwd_part, void* ws, size_t ws_bytes, void* stream);
/* dw[n,k] = sum_m dy[m,n] a(x[m,k]) (fp32);  dbias[n] = sum_m dy[m,n] (or NULL) */
size_t tsii_bf16_pw_bwd_dw_ws_bytes(int64_t m, int n, int k);
int tsii_bf16_pw_bwd_dw(const uint16_t* dy, const uint16_t* x, int64_t m, int n, int k,
                        const float* in_scale, const float* in_shift, int in_act, float in_slope,
                        float* dw, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* ---- dense k x k convolutions, groups == 1, as implicit GEMM (w: reference layout [cout,cin,kh,kw], fp32) ------------- */
size_t tsii_bf16_dense_ws_bytes(int cin, int cout, int kh, int kw);
int tsii_bf16_dense_fwd(const uint16_t* x, const float* w, const float* bias, int n, int h, int wd, int cin, int cout,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                        float* stat_part /* [tsii_bf16_stat_rows(n*ho*wo)][4][cout] or NULL */, uint16_t* y,
                        void* ws, size_t ws_bytes, void* stream);
int tsii_bf16_dense_bwd_dx(const uint16_t* dy, const float* w, int n, int h, int wd, int cin, int cout,
                           int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                           uint16_t* dx, void* ws, size_t ws_bytes, void* stream);
size_t tsii_bf16_dense_bwd_dw_ws_bytes(int n, int ho, int wo, int cin, int cout, int kh, int kw);
int tsii_bf16_dense_bwd_dw(const uint16_t* dy, const uint16_t* x, int n, int h, int wd, int cin, int cout,
                           int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                           float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- depth-wise 3x3 convolutions (Conv_block(groups = C), models/BaseModels.py:105-127) and stride-1 average pools -------
 * w: reference layout [c,1,3,3] fp32.  Geometry: 3x3, stride 1 with any dilation or stride 2 with dilation 1, square padding.
 * in_scale / in_shift / stat_part: K6b as for the fp32 entry points (the virtual activation a = act(in_scale*x + in_shift) is
 * rounded to bf16 like a stored one; zero padding pads a).  tsii_bf16_dw_stat_rows: rows of stat_part [rows][4][c] (0: geometry
 * not supported).  The forward and the stride-1 dX are one kernel (the adjoint is the same stencil with flipped taps). */
int64_t tsii_bf16_dw_stat_rows(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw);
int tsii_bf16_dw_fwd(const uint16_t* x, const float* w, const float* bias, int n, int h, int wd, int c,
                     int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                     const float* in_scale, const float* in_shift, int in_act, float in_slope,
                     float* stat_part, uint16_t* y, void* stream);
/* K6c: with bn_y (the raw [n,h,wd,c] tensor this conv consumed through its load-time BatchNorm) the kernel also leaves
 * bwd_part[tsii_bf16_dw_bwd_stat_rows(..)][2][c]; rows == 0 (stride 2): pass NULL and run tsii_bf16_bn_act_bwd instead */
int64_t tsii_bf16_dw_bwd_stat_rows(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);
int tsii_bf16_dw_bwd_dx(const uint16_t* dy, const float* w, int n, int h, int wd, int c,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                        const uint16_t* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                        const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                        uint16_t* dx, float* bwd_part, void* stream);
size_t tsii_bf16_dw_bwd_dw_ws_bytes(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw);
int tsii_bf16_dw_bwd_dw(const uint16_t* dy, const uint16_t* x, int n, int h, int wd, int c,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                        const float* in_scale, const float* in_shift, int in_act, float in_slope,
                        float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream);
/* nn.AvgPool2d(k, stride 1, padding (k-1)/2), count_include_pad (models/common.py:62-68), k in {3, 5, 9}; the operator is
 * its own adjoint: the backward pass is the same call on the gradient */
int tsii_bf16_avgpool(const uint16_t* x, int n, int h, int wd, int c, int k, uint16_t* y, void* stream);

/* ---- streaming kernels ------------------------------------------------------------------------------------------------
 * BatchNorm statistics of a bf16 [m,c] tensor as partial rows in the stat_part layout ([tsii_bf16_bn_stat_rows(m,c)][4][c]);
 * tsii_bn_finalize turns them into mean / var / running statistics / (scale, shift) like the conv-emitted partials. */
int64_t tsii_bf16_bn_stat_rows(int64_t m, int c);
int tsii_bf16_bn_stats(const uint16_t* y, int64_t m, int c, float* stat_part, void* stream);
/* out = act(scale*y + shift) (+ residual): the materialised form of a BatchNorm(+activation) (models/BaseModels.py:95-99)
 * from the (scale, shift) pair of tsii_bn_finalize / tsii_bn_scale_shift, i.e. exactly what a load-time consumer computes */
int tsii_bf16_bn_act_fwd(const uint16_t* y, int64_t m, int c, const float* scale, const float* shift, int act, float slope,
                         const uint16_t* residual, uint16_t* out, void* stream);
/* BatchNorm(+act) backward: dy (bf16), dgamma / dbeta (fp32).  bwd_part NULL: the kernel takes its two reductions itself;
 * else [rows][2][c] partials left by the consumer's dX kernel (K6c).  ws: tsii_bf16_bn_ws_bytes(m, c). */
size_t tsii_bf16_bn_ws_bytes(int64_t m, int c);
int tsii_bf16_bn_act_bwd(const uint16_t* dout, const uint16_t* y, int64_t m, int c, const float* mean, const float* var,
                         const float* gamma, const float* beta, float eps, int act, float slope, int training,
                         const float* bwd_part, int64_t rows, uint16_t* dy, float* dgamma, float* dbeta,
                         void* ws, size_t ws_bytes, void* stream);
/* out = act(a + b); dx = dout * act'(x)   (models/Xception.py:44, models/MobileNetV2.py:146-147, models/common.py:156) */
int tsii_bf16_add_act_fwd(const uint16_t* a, const uint16_t* b, int64_t numel, int act, float slope, uint16_t* out, void* stream);
int tsii_bf16_act_bwd(const uint16_t* dout, const uint16_t* x, int64_t numel, int act, float slope, uint16_t* dx, void* stream);
/* channel concat / slice (tsii_copy_channels), bilinear up-sampling (tsii_bilinear_up_*) on bf16 tensors */
int tsii_bf16_copy_channels(uint16_t* big, int64_t m, int cbig, int coff, uint16_t* small_, int csmall, int to_dst, void* stream);
int tsii_bf16_bilinear_up_fwd(const uint16_t* x, int n, int h, int w, int c, int scale, uint16_t* y, void* stream);
int tsii_bf16_bilinear_up_bwd(const uint16_t* dy, int n, int h, int w, int c, int scale, uint16_t* dx, void* stream);
/* the fp32 <-> bf16 boundary: the 3-channel image enters through the stem's space-to-depth rearrangement (tsii_stem_s2d with
 * the channels of each phase padded to 4: out [n,(h+2pad)/2,(w+2pad)/2,16], channel order (row phase, column phase, c4));
 * the logits leave as channel `ch` of a head padded to 8 channels; plain casts for everything else */
int tsii_bf16_stem_s2d(const float* x, int n, int h, int w, int c, int pad, uint16_t* out, void* stream);
int tsii_bf16_from_f32(const float* src, int64_t numel, uint16_t* dst, void* stream);
int tsii_bf16_to_f32(const uint16_t* src, int64_t numel, float* dst, void* stream);
int tsii_bf16_channel_to_f32(const uint16_t* src, int64_t m, int c, int ch, float* dst, void* stream);
int tsii_bf16_channel_from_f32(const float* src, int64_t m, int c, int ch, uint16_t* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSII_HIP_H */
