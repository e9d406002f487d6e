// K2 (round 4): the stride-1 / dilation-1 marching strip, rebuilt around the instruction budget.
//
// The round-3 strip kernel (dw_strip_kernel<1, 1, *> in dwconv.hip) was ISSUE-bound, not HBM-bound: rocprofv3 put its waves at
// 45 % active-issue with two waves per SIMD, and its main loop spent ~190 vector instructions per 4-channel output pixel -- ring
// rows taken `% 10` per tap row, the v_pk_fma_f32 pairs hipcc chose needing two v_mov per LDS read, one IEEE division per pixel
// in each of the 8 threads sharing it, 64-bit address arithmetic and exec-mask branches around every load.  This kernel does the
// same arithmetic in the same order (outputs are bit-identical) with a third of that:
//   * TWO plain LDS buffers of 10 input rows instead of a ring: every LDS address is a per-thread base + an immediate offset.
//     The two rows a step shares with the next one are copied LDS -> LDS into the other buffer while the step runs, and the next
//     slab is committed into that other buffer too, so a step needs ONE barrier (nothing ever writes the buffer being read).
//   * a thread owns 4 vertically adjacent pixels of a column: 6 input rows x 3 columns = 18 ds_read_b128 feed 144 FMAs
//     (the old mapping read 36), written as explicit float2 pairs on the natural register pairs of the 16-byte reads.
//   * per-pixel planes are reduced by the threads that fetch them (1 / denom once per pixel, as a (1/denom, keep) pair).
//   * slab loads are wave-uniform running pointers + 32-bit offsets computed once; interior steps load without bounds arithmetic.
//   * per-channel constants other than the 9 x 4 weights (BatchNorm scale / shift, bias, the K6c constants) sit in LDS and are
//     read where they are used: 3 waves per SIMD.
//   * BatchNorm statistics (K6b) use the thread's first output as pivot without a per-pixel select; the K6c reductions run
//     after the step's stores, when the BatchNorm-input loads issued at the top of the step have landed.
// Included by dwconv.hip (after DtGeom / DwBN / DwBnBwd).
#pragma once

// a value the compiler must have materialised at this point (keeps the FMAs that produce it from being sunk); no-op on the test emulator
#ifndef TSII_PIN_F2
#define TSII_PIN_F2(x) asm volatile("" : "+v"(x))
#endif

// A/B build knobs (tools/variants/build_variant.py); the defaults are what measured best on the MI355X
#ifndef LS_NT_STORE
#define LS_NT_STORE 1          // non-temporal output stores (A/B on the 256^2 x 384 layer: fwd_bn 1.269 -> 1.214 ms)
#endif
#ifndef LS_NT_LOAD
#define LS_NT_LOAD 0           // non-temporal slab loads
#endif
#ifndef LS_ORDER
#define LS_ORDER 1             // block order: 0 channel block fastest, 1 strip fastest, then channel block, chunk, image (fwd 1.227 -> 1.160 ms)
#endif
#ifndef LS_WAVES_FUSED
#define LS_WAVES_FUSED 3       // waves per SIMD the K6b / K6c forms are compiled for
#endif

namespace tsii {

typedef float f32x2 __attribute__((ext_vector_type(2)));

static constexpr int LS_R = 8, LS_TW = 16, LS_PW = 18, LS_ROWS = 10, LS_CB = 32, LS_PF = 5;
static constexpr int LS_PIXB = LS_CB * 4;                    // bytes of one staged pixel
static constexpr int LS_BUFB = LS_ROWS * LS_PW * LS_PIXB;    // bytes per buffer (23040)
static constexpr int LS_NPX = LS_R * LS_TW;

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 max2(f32x2 a, f32x2 b) { return f32x2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__device__ __forceinline__ f32x2 min2(f32x2 a, f32x2 b) { return f32x2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
__device__ __forceinline__ f32x4 cat4(f32x2 a, f32x2 b) { return f32x4{a.x, a.y, b.x, b.y}; }

// MODE 0: plain; 1: forward with BatchNorm on load and / or statistics partials (K6b); 2: dX feeding a BatchNorm backward (K6c);
// 3 (K6d, round 6): MODE 2 that ALSO takes the layer's weight gradient.  The dX pass already holds both operands of dW on chip:
// the staged slab is G = dy * inv with its halo, and the raw BatchNorm input y it reads for the K6c reductions at its own output
// pixels q gives the convolution's input a(q) = act(BatchNorm(y(q))) * rmask(q) for three more instructions per element --
//     dW[t] = sum_q a(q) G(q + pad - t)        dX(q) = rmask(q) sum_t W[t] G(q + pad - t)
// walk the same 3x3 window of G around q.  A second sweep over the step's 18 LDS reads (after the stores, when y has landed)
// accumulates 9 taps x 4 channels per thread over the whole strip chunk; one partial row [9][C] per block in `stats`, combined
// by dw_reduce_kernel.  The separate weight-gradient kernel (dw_strip_dw_kernel<1, 1>: dy and y streamed once more, 6.4 GB on
// the 32 x 256^2 x 384 layer) is not launched.  36 more accumulators: 2 waves per SIMD.
// 4 (K6e, round 6): MODE 3 whose staged slab is FORMED from the gradient w.r.t. the layer's OUTPUT activation: the BatchNorm
// that follows the layer (its output y2 -> act(BatchNorm(y2))) is differentiated while the slab is staged,
//     G = (dz - k1 - xhat k2) gamma istd * inv,   dz = da * act'(z),  xhat = (y2 - mu) istd,  z = xhat gamma + beta
// from the two tensors (da, y2) and the per-channel table coef[6][C] = (mu, istd, gamma, beta, k1, k2) the reduction kernel
// leaves (bn.hip: bn_bwd_final_kernel) -- the stand-alone apply pass (tsii_bn_act_bwd_pre: read da, read y2, write dy; 9.7 GB
// on the 32 x 256^2 x 384 layer) and this kernel's read of its result are replaced by this kernel reading da and y2.
// The operands ride in `ib`: ib.sc = y2 (same grid and pitch as `in` = da), ib.sh = coef, ib.neg / ib.hi = that activation's.
// DXE: dX epilogue (out = post_mul != 0 ? acc * post_mul : 0) instead of the forward one (acc / denom + bias, zero where keep == 0).
// PRE: the staged input is multiplied by a per-pixel plane (mask for forward, 1 / count for dX).
// PH (round 6): DILATION d > 1 BY PHASES.  With dilation d every tap of output pixel (oy, ox) lies d pixels apart, so the outputs with
// oy = py (mod d), ox = px (mod d) together with the inputs they read form an ordinary dilation-1 3x3 convolution on the sub-image
// of every d-th row and column -- d^2 independent ones per image.  A block takes a strip chunk of ONE phase: the same kernel, on a
// "virtual" image whose pixel (r, c) is the real pixel (p_y + d r, p_x + d c), i.e. every pixel offset of the d = 1 kernel times d
// from the phase's first pixel (one pixel of a 32-channel block is a 128-byte segment either way).  No halo beyond the one
// sub-pixel of the 3x3 window: a dilation-8 layer reads 18 / 16 of its input like a dilation-1 layer, where the ring of 8 + 2 d
// rows x (16 + 2 d) columns of the round-3 strip kernel read 2x (64^2 x 1920 at d = 8: 1.7 TB/s).  The grid carries image x phase
// as its slowest index (nv = n d^2 + phase: the partial-row layout of the fused forms follows it, plan_strip_phased).
// H16 (round 6): bf16 ACTIVATION STORAGE on the same kernel (bf16_dw.hip's contract: read bf16, compute in fp32, ONE rounding per
// stored value, statistics over the rounded values): `in`, `out` and `bb.y` point at bf16 tensors, a thread moves its 4 channels as
// 8 bytes, the staged slab in LDS is fp32 (an activation the producer's BatchNorm forms on load is rounded to bf16 first -- where it
// would have been stored), `wT` is the reference layout [C][9] (the bf16 entry points carry no workspace for a transposed copy).
template <int MODE, bool DXE, bool PRE, bool PH = false, bool H16 = false>
__global__ __launch_bounds__(256, MODE == 0 ? 3 : (MODE >= 3 ? 2 : LS_WAVES_FUSED)) void dw_lean_kernel(
    const float* __restrict__ in, const float* __restrict__ pre, const float* __restrict__ wT, const float* __restrict__ bias,
    const float* __restrict__ denom, const float* __restrict__ keep, const float* __restrict__ post_mul, DtGeom g, int chunk_rows,
    unsigned strips_x, unsigned chunks_y, unsigned cblocks, DwBN ib, float* __restrict__ stats, DwBnBwd bb,
    float* __restrict__ out) {
    constexpr bool FUSED = (MODE == 1), BNB = (MODE >= 2), DWG = (MODE >= 3), APL = (MODE == 4);
    static_assert(!DWG || (!PH && !H16), "K6d: the fp32 dilation-1 strips only");
    constexpr unsigned ES = H16 ? 2u : 4u;                   // bytes per activation element
    static_assert(!H16 || !PRE, "bf16 activation storage carries no mask planes");
    static_assert(!BNB || DXE, "K6c rides on the dX epilogue");
    static_assert(!FUSED || !DXE, "K6b rides on the forward epilogue");
    __shared__ __attribute__((aligned(16))) float lbuf[2 * LS_BUFB / 4];
    __shared__ __attribute__((aligned(8))) float lplanes[2][LS_NPX][2];
    __shared__ __attribute__((aligned(16))) float lconst[4][LS_CB];   // K6b: scale, shift, bias | K6c: mean, 1/sigma, gamma, beta
    __shared__ __attribute__((aligned(16))) float lapl[APL ? 6 : 1][LS_CB];   // K6e: mu, istd, gamma, beta, k1, k2 of the folded BatchNorm
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    unsigned cb, sx;
    if (LS_ORDER == 1) { sx = b % strips_x; b /= strips_x; cb = b % cblocks; b /= cblocks; }
    else { cb = b % cblocks; b /= cblocks; sx = b % strips_x; b /= strips_x; }
    const unsigned cy = b % chunks_y;
    const int64_t nv = b / chunks_y;                          // image (x phase): index of the block's partial row
    int64_t n = nv;
    // the phase's view: D = pixel pitch, (vhin, vwin) / (vhout, vwout) the sub-image sizes, (pvh, pvw) its padding,
    // ibase0 / obase0 = first pixel of the phase in the input / output grid (pixel index inside the image)
    int D = 1, vhin = g.hin, vwin = g.win, vhout = g.hout, vwout = g.wout, pvh = g.pad_h, pvw = g.pad_w;
    int ipix0 = 0, opix00 = 0;
    bool have_in = true;
    if (PH) {
        D = g.d;
        const int dd = D * D;
        n = nv / dd;
        const int ph = (int)(nv - n * dd);
        const int py_o = ph / D, px_o = ph - py_o * D;
        // input row of (virtual output row r, tap ky) = py_o + D r - pad + D ky: the input phase and the padding of the view
        const int qy = py_o - g.pad_h, qx = px_o - g.pad_w;
        int py_i = qy % D, px_i = qx % D;
        py_i = py_i < 0 ? py_i + D : py_i; px_i = px_i < 0 ? px_i + D : px_i;
        pvh = (py_i - qy) / D; pvw = (px_i - qx) / D;
        vhout = g.hout > py_o ? (g.hout - py_o + D - 1) / D : 0;
        vwout = g.wout > px_o ? (g.wout - px_o + D - 1) / D : 0;
        vhin = g.hin > py_i ? (g.hin - py_i + D - 1) / D : 0;
        vwin = g.win > px_i ? (g.win - px_i + D - 1) / D : 0;
        have_in = vhin > 0 && vwin > 0;                        // (a phase without input pixels: every tap is padding)
        if (!have_in) { py_i = 0; px_i = 0; vhin = 1; vwin = 1; }   // clamped loads stay inside the image; their values are dropped
        ipix0 = py_i * g.win + px_i;
        opix00 = py_o * g.wout + px_o;
    }
    const unsigned Du = (unsigned)D;
    const int t = threadIdx.x;
    const int cg = t & 7, lane = t >> 3;
    const int C = g.c;
    const int c0 = (int)cb * LS_CB + cg * 4;
    const bool cok = c0 < C;
    const int oy_beg = (int)cy * chunk_rows;
    const int oy_end = oy_beg + chunk_rows < vhout ? oy_beg + chunk_rows : vhout;
    const int ox0 = (int)sx * LS_TW;
    const int iy_base = oy_beg - pvh, ix0 = ox0 - pvw;        // input row of buffer row 0 at step 0 / input column of slab column 0
    const int nsteps = (oy_end - oy_beg + LS_R - 1) / LS_R;
    const bool col_interior = have_in && ix0 >= 0 && ix0 + LS_PW <= vwin;      // block-uniform
    const float bn_neg = FUSED && ib.sc != nullptr ? ib.neg : 1.f;   // no producer BatchNorm: the identity (scale 1, shift 0, neg 1), exact
    const bool hi_finite = FUSED && ib.sc != nullptr && ib.hi < __builtin_huge_valf();
    if (PH && (nsteps <= 0 || ox0 >= vwout)) {
        // a phase with fewer rows / columns than the plan's largest: nothing to compute, but the fused forms' partial rows exist
        if (t < LS_CB && (int)cb * LS_CB + t < C) {
            const int64_t prow = (nv * chunks_y + cy) * strips_x + sx;
            if (BNB) { bb.part[(prow * 2 + 0) * C + (int)cb * LS_CB + t] = 0.f; bb.part[(prow * 2 + 1) * C + (int)cb * LS_CB + t] = 0.f; }
            else if (FUSED && stats != nullptr) {
                float* sp = stats + prow * 4 * C + (int)cb * LS_CB + t;
                sp[0] = 0.f; sp[C] = 0.f; sp[2 * (int64_t)C] = 0.f; sp[3 * (int64_t)C] = 0.f;
            }
        }
        return;
    }

    if (t < LS_CB) {                                          // per-channel constants of this block's 32 channels
        const int ch = (int)cb * LS_CB + t;
        const bool ok = ch < C;
        if (APL) {
            const float* cf = ib.sh;
#pragma unroll
            for (int j = 0; j < 6; ++j) lapl[j][t] = ok ? cf[(int64_t)j * C + ch] : 0.f;
        }
        if (BNB) {
            lconst[0][t] = ok ? bb.mean[ch] : 0.f;
            lconst[1][t] = ok ? 1.0f / sqrtf(bb.var[ch] + bb.eps) : 0.f;
            lconst[2][t] = ok ? bb.gamma[ch] : 0.f;
            lconst[3][t] = ok ? bb.beta[ch] : 0.f;
        } else {
            lconst[0][t] = (FUSED && ib.sc != nullptr && ok) ? ib.sc[ch] : 1.f;
            lconst[1][t] = (FUSED && ib.sc != nullptr && ok) ? ib.sh[ch] : 0.f;
            lconst[2][t] = (!DXE && bias != nullptr && ok) ? bias[ch] : 0.f;
        }
    }
    // K6e keeps the 9 x 4 weights in LDS and walks the window tap row by tap row (3 weight vectors and one slab row live at a time
    // instead of 9 + 9: the second input stream's registers have to come from somewhere at 2 waves per SIMD); everything else
    // holds them in registers
    constexpr bool WL = APL;
    __shared__ __attribute__((aligned(16))) float lw[WL ? 9 : 1][LS_CB];
    f32x4 w[WL ? 1 : 9];
#pragma unroll
    for (int k = 0; k < (WL ? 1 : 9); ++k) w[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (WL) {
        for (int j = t; j < 9 * LS_CB; j += 256) {
            const int k = j / LS_CB, ch = (int)cb * LS_CB + j % LS_CB;
            lw[k][j % LS_CB] = ch < C ? wT[(g.flip ? 8 - k : k) * C + ch] : 0.f;
        }
    } else if (cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int kk = g.flip ? 8 - k : k;
            if (H16) w[k] = f32x4{wT[(c0 + 0) * 9 + kk], wT[(c0 + 1) * 9 + kk], wT[(c0 + 2) * 9 + kk], wT[(c0 + 3) * 9 + kk]};
            else w[k] = *reinterpret_cast<const f32x4*>(wT + kk * C + c0);
        }
    }

    // ---- slab staging: 8 rows x 18 pixels per step; slab pixel p = lane + 32 i of thread t sits at byte (t + 256 i) * 16 ----
    // EVERY global load below is unconditional (threads without a valid item load a clamped, valid address and drop the value)
    // and so is its first use: a load inside an exec-masked region whose consumer sits in another one leaves hipcc's waitcnt
    // scoreboard with a pending entry on the skipping path, and the next load into that register then waits vmcnt(0) -- for
    // the loads issued just before it.
    unsigned poff[LS_PF];                  // byte offset of the item's pixel in a per-pixel plane, from the slab's first pixel
    unsigned rowpk = 0, pxpk = 0;
#pragma unroll
    for (int i = 0; i < LS_PF; ++i) {
        const int pr = lane + 32 * i;
        const int p = pr < LS_R * LS_PW ? pr : LS_R * LS_PW - 1;   // item 4 of the pixel lanes >= 16: a second load of the slab's last pixel
        const int row = p / LS_PW, px = p - row * LS_PW;
        poff[i] = (unsigned)(row * g.win + px) * 4u * Du;
        rowpk |= (unsigned)row << (4 * i);
        pxpk |= (unsigned)px << (5 * i);
    }
    const bool item4 = lane < LS_R * LS_PW - 128;             // item 4 exists for the first 16 pixel lanes only
    const unsigned c0b = (unsigned)(cok ? c0 : C - 4) * ES;   // channel offset for loads (clamped: the channel tail of the last block)
    // Offsets pass through an opaque copy right before their use: the zero-extension then sits next to the access and hipcc
    // selects the `global_load v, v_off, s[base]` form instead of 64-bit vector address pairs
    auto opq = [](unsigned o) { TSII_OPAQUE_U32(o); return o; };
    // 4 channels of an activation tensor at byte address p: 16 bytes of fp32, or 8 bytes of bf16 widened (exact)
    auto ld4 = [](const char* p) {
        if constexpr (H16) {
            const unsigned long long u = *reinterpret_cast<const unsigned long long*>(p);
            const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
            return f32x4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                         __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
        } else {
            return *reinterpret_cast<const f32x4*>(p);
        }
    };
    // byte offset of 4 channels from a plane byte offset po (= 4 x pixel index): po x C elements of ES bytes
    auto aoff = [&](unsigned po) { return H16 ? (__umul24(po, (unsigned)C) >> 1) : __umul24(po, (unsigned)C); };
    // round-to-nearest-even through bf16 (v_cvt_pk_bf16_f32), as a stored value would be
    auto rnd2 = [](f32x2 v) {
        typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
        const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b16x2));
        return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    };

    f32x4 pf[LS_PF];
    f32x4 pf2[APL ? LS_PF : 1];                                // K6e: the folded BatchNorm's raw input at the same pixels
    float pm[LS_PF];
    unsigned vmask = 0;                                        // edge steps: bit i = item i lies inside the image
    const int64_t img_pix = n * g.hin * (int64_t)g.win + ipix0;
    const char* const ibase = reinterpret_cast<const char*>(in) + img_pix * C * ES;      // pixel (0, 0) of this image (of its phase)
    const int64_t y2d = APL ? reinterpret_cast<const char*>(ib.sc) - reinterpret_cast<const char*>(in) : 0;   // K6e: y2 - da, in bytes (same layout)
    const char* const ipre = reinterpret_cast<const char*>(pre + img_pix);
    // running pointers to the first pixel of the NEXT slab to fetch (may point outside the tensor; only used by interior steps)
    int iyb = iy_base + 2;
    const char* sb = reinterpret_cast<const char*>(in) + (img_pix + ((int64_t)iyb * g.win + ix0) * D) * C * ES;
    const char* pb = reinterpret_cast<const char*>(pre + (img_pix + ((int64_t)iyb * g.win + ix0) * D));
    const int64_t sb_step = (int64_t)LS_R * g.win * C * ES * D, pb_step = (int64_t)LS_R * g.win * 4 * D;
    auto fetch = [&]() {                                      // global -> registers, slab rows iyb .. iyb + 7
        const bool interior = col_interior && iyb >= 0 && iyb + LS_R <= vhin;
        unsigned po[LS_PF];
        vmask = 31u;
#pragma unroll
        for (int i = 0; i < LS_PF; ++i) po[i] = poff[i];
        if (!interior) {                                       // edge step: offsets of the clamped pixel from the image's first pixel
            vmask = 0;
#pragma unroll
            for (int i = 0; i < LS_PF; ++i) {
                const int iy = iyb + (int)((rowpk >> (4 * i)) & 15u), ix = ix0 + (int)((pxpk >> (5 * i)) & 31u);
                if (have_in && (unsigned)iy < (unsigned)vhin && (unsigned)ix < (unsigned)vwin) vmask |= 1u << i;
                const int iyc = iy < 0 ? 0 : (iy >= vhin ? vhin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= vwin ? vwin - 1 : ix);
                po[i] = (unsigned)(iyc * g.win + ixc) * 4u * Du;
            }
        }
        const char* const ab = interior ? sb : ibase;
        const char* const mb = interior ? pb : ipre;
#pragma unroll
        for (int i = 0; i < LS_PF; ++i) {
            pf[i] = (LS_NT_LOAD && !H16) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ab + opq(aoff(po[i]) + c0b)))
                                         : ld4(ab + opq(aoff(po[i]) + c0b));
            if (APL) pf2[i] = ld4(ab + y2d + opq(aoff(po[i]) + c0b));
            pm[i] = PRE ? *reinterpret_cast<const float*>(mb + opq(po[i])) : 1.f;
        }
        iyb += LS_R; sb += sb_step; pb += pb_step;
    };
    // BatchNorm + activation of the producer (K6b), then the per-pixel plane; items outside the image become exact zeros (never
    // `x * 0`: the clamped address may hold NaN / Inf) -- zero padding pads the ACTIVATED tensor
    char* const lthr = reinterpret_cast<char*>(lbuf) + t * 16;
    const char* const cthr = reinterpret_cast<const char*>(&lconst[0][0]) + cg * 16;
    // K6e: the folded BatchNorm's backward on one element quad: da, y2 -> dy (constants from LDS at the point of use)
    const char* const athr = reinterpret_cast<const char*>(&lapl[0][0]) + cg * 16;
    auto bn_apply = [&](f32x4 da, f32x4 y2) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(athr), is = *reinterpret_cast<const f32x4*>(athr + LS_PIXB);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(athr + 2 * LS_PIXB), be = *reinterpret_cast<const f32x4*>(athr + 3 * LS_PIXB);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(athr + 4 * LS_PIXB), k2 = *reinterpret_cast<const f32x4*>(athr + 5 * LS_PIXB);
        const f32x2 h0 = (y2.xy - mu.xy) * is.xy, h1 = (y2.zw - mu.zw) * is.zw;
        const f32x2 z0 = fma2(h0, ga.xy, be.xy), z1 = fma2(h1, ga.zw, be.zw);
        f32x2 d0 = da.xy, d1 = da.zw;
        d0.x *= (z0.x > 0.f && z0.x < ib.hi) ? 1.f : (z0.x > 0.f ? 0.f : ib.neg);
        d0.y *= (z0.y > 0.f && z0.y < ib.hi) ? 1.f : (z0.y > 0.f ? 0.f : ib.neg);
        d1.x *= (z1.x > 0.f && z1.x < ib.hi) ? 1.f : (z1.x > 0.f ? 0.f : ib.neg);
        d1.y *= (z1.y > 0.f && z1.y < ib.hi) ? 1.f : (z1.y > 0.f ? 0.f : ib.neg);
        // the stand-alone pass's order: ((dz - k1) - xhat k2) * gamma * istd  (bn_bwd_apply_kernel)
        d0 = (d0 - k1.xy) - h0 * k2.xy; d1 = (d1 - k1.zw) - h1 * k2.zw;
        return cat4((d0 * ga.xy) * is.xy, (d1 * ga.zw) * is.zw);
    };
    auto stage = [&](f32x4 v, float m, bool inside, const f32x4& isc, const f32x4& ish) {
        if (FUSED) {
            f32x2 z0 = fma2(v.xy, isc.xy, ish.xy), z1 = fma2(v.zw, isc.zw, ish.zw);
            z0 = max2(z0, z0 * bn_neg); z1 = max2(z1, z1 * bn_neg);
            if (hi_finite) { z0 = min2(z0, f32x2{ib.hi, ib.hi}); z1 = min2(z1, f32x2{ib.hi, ib.hi}); }
            if (H16 && ib.sc != nullptr) { z0 = rnd2(z0); z1 = rnd2(z1); }      // the virtual activation is a bf16 tensor
            v = cat4(z0, z1);
        }
        v *= m;
        if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
        return v;
    };
    // registers -> buffer `which` rows 2..9 (threads of a channel tail write their never-read slots too)
    auto commit_slab = [&](int which, const f32x4& isc, const f32x4& ish) {
        char* const T = lthr + which * LS_BUFB + 2 * LS_PW * LS_PIXB;
        if (vmask == 31u) {                                    // interior step (wave-uniform in practice; any mix is handled below)
#pragma unroll
            for (int i = 0; i < LS_PF; ++i) {
                const f32x4 v = stage(APL ? bn_apply(pf[i], pf2[APL ? i : 0]) : pf[i], pm[i], true, isc, ish);
                if (i < 4 || item4) *reinterpret_cast<f32x4*>(T + 256 * i * 16) = v;
                if (APL) __builtin_amdgcn_sched_barrier(0);        // K6e: one item's transform at a time (VGPR budget)
            }
        } else {
#pragma unroll
            for (int i = 0; i < LS_PF; ++i) {
                const f32x4 v = stage(APL ? bn_apply(pf[i], pf2[APL ? i : 0]) : pf[i], pm[i], (vmask >> i) & 1u, isc, ish);
                if (i < 4 || item4) *reinterpret_cast<f32x4*>(T + 256 * i * 16) = v;
            }
        }
    };
    // rows 0..1 of buffer `which` = rows 8..9 of the other buffer (which nothing writes during this step)
    auto commit = [&](int which) {
        char* const T = lthr + which * LS_BUFB;
        const char* const S = lthr + (which ^ 1) * LS_BUFB + 8 * LS_PW * LS_PIXB;
        const f32x4 c0v = *reinterpret_cast<const f32x4*>(S);
        f32x4 c1v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < 2 * LS_PW * 8 - 256) c1v = *reinterpret_cast<const f32x4*>(S + 256 * 16);
        f32x4 isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f};
        if (FUSED) { isc = *reinterpret_cast<const f32x4*>(cthr); ish = *reinterpret_cast<const f32x4*>(cthr + LS_PIXB); }
        *reinterpret_cast<f32x4*>(T) = c0v;
        if (t < 2 * LS_PW * 8 - 256) *reinterpret_cast<f32x4*>(T + 256 * 16) = c1v;
        commit_slab(which, isc, ish);
    };
    // per-pixel planes of step s (one output pixel per thread, threads 128.. repeat 0..127), reduced to what the epilogue multiplies by
    float pl0 = 1.f, pl1 = 1.f;
    auto fetch_planes = [&](int s) {
        const int tp = t & (LS_NPX - 1);
        const int oy = oy_beg + LS_R * s + tp / LS_TW, ox = ox0 + tp % LS_TW;
        const bool ok = oy < oy_end && ox < vwout;
        const int64_t q = n * g.hout * (int64_t)g.wout + opix00 + ((int64_t)(ok ? oy : oy_beg) * g.wout + (ok ? ox : ox0)) * D;
        if (DXE) {
            pl0 = post_mul != nullptr ? post_mul[q] : 1.f;
        } else {
            pl0 = denom != nullptr ? denom[q] : 1.f;
            pl1 = keep != nullptr ? keep[q] : 1.f;
        }
    };
    auto commit_planes = [&](int s) {
        // one IEEE division per pixel, then multiplies (<= 1 ulp from the reference's division)
        const float r0 = DXE ? pl0 : 1.0f / pl0, r1 = pl1;
        if (t < LS_NPX) {
            lplanes[s & 1][t][0] = r0;
            lplanes[s & 1][t][1] = r1;
        }
    };

    // ---- prologue: input rows 0..1 of the first step -> buffer 0 rows 0..1, slab 0 -> rows 2..9 ----------------------------
    fetch_planes(0);
    __syncthreads();                                           // lconst
    {
        const f32x4 isc = *reinterpret_cast<const f32x4*>(cthr), ish = *reinterpret_cast<const f32x4*>(cthr + LS_PIXB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pr = lane + 32 * i;
            const int p = pr < 2 * LS_PW ? pr : 2 * LS_PW - 1;
            const int row = p / LS_PW, px = p - row * LS_PW;
            const int iy = iy_base + row, ix = ix0 + px;
            const bool inside = have_in && (unsigned)iy < (unsigned)vhin && (unsigned)ix < (unsigned)vwin;
            const int iyc = iy < 0 ? 0 : (iy >= vhin ? vhin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= vwin ? vwin - 1 : ix);
            const unsigned q = (unsigned)(iyc * g.win + ixc) * 4u * Du;
            f32x4 v = ld4(ibase + opq(aoff(q) + c0b));
            if (APL) v = bn_apply(v, ld4(ibase + y2d + opq(aoff(q) + c0b)));
            const float m = PRE ? *reinterpret_cast<const float*>(ipre + opq(q)) : 1.f;
            const f32x4 sv = stage(v, m, inside, isc, ish);
            if (pr < 2 * LS_PW) *reinterpret_cast<f32x4*>(lthr + 256 * i * 16) = sv;
        }
        fetch();
        commit_slab(0, isc, ish);
    }
    commit_planes(0);
    if (nsteps > 1) { fetch(); fetch_planes(1); }              // slab 1: in flight across the barrier and the first step's compute
    __syncthreads();

    // this thread's output pixels: column tx, rows 4 th .. 4 th + 3 of the step
    const int tx = lane & 15, th = lane >> 4;
    const bool xok = cok && ox0 + tx < vwout;
    const unsigned orow = (unsigned)g.wout * (unsigned)C * ES * Du;
    const unsigned ocol = ((unsigned)tx * Du * (unsigned)C + (unsigned)c0) * ES;
    const unsigned ocol_ld = ((unsigned)(ox0 + tx < vwout ? tx : vwout - 1 - ox0) * Du * (unsigned)C) * ES + c0b;   // clamped, for loads
    const char* const rthr = reinterpret_cast<const char*>(lbuf) + ((4 * th) * LS_PW + tx) * LS_PIXB + cg * 16;
    const int64_t opix0 = n * g.hout * (int64_t)g.wout + opix00 + ((int64_t)oy_beg * g.wout + ox0) * D;
    char* ob = reinterpret_cast<char*>(out) + opix0 * C * ES;                  // running: first output pixel of the step
    const char* const yb = reinterpret_cast<const char*>(bb.y) + opix0 * C * ES;
    const int64_t ob_step = (int64_t)LS_R * g.wout * C * ES * D;

    f32x4 P = {0.f, 0.f, 0.f, 0.f};                    // K6b: thread-local pivot = its first output
    f32x2 va[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // K6b: s1.xy s1.zw s2.xy s2.zw | K6c: sum dz, sum dz*xhat

    f32x4 dwa[DWG ? 9 : 1];                              // K6d: weight-gradient accumulators, tap (ky, kx) of the (flipped) window
#pragma unroll
    for (int k = 0; k < (DWG ? 9 : 1); ++k) dwa[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K6c: raw BatchNorm input at this thread's output pixels of step s (clamped rows / columns: the loads are unconditional),
    // requested one step ahead like the slab and used after the step's stores
    f32x4 yv[4];
    auto fetch_y = [&](int s) {
        const int rows_left = oy_end - (oy_beg + LS_R * s);
        const char* const ybs = yb + (int64_t)s * ob_step;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ry = 4 * th + k < rows_left ? 4 * th + k : rows_left - 1;
            yv[k] = ld4(ybs + opq((unsigned)ry * orow + ocol_ld));
        }
    };
    if (BNB) fetch_y(0);
    // One step.  `more`: another one follows -- its slab and planes were requested at the end of the previous step (right after
    // the registers they land in were committed: a block always has a slab in flight, also while it waits at the barrier) and
    // are committed after this step's compute; `more2`: the slab after that is requested then.
    auto step = [&](int s, bool more, bool more2) {
        const int rows_left = oy_end - (oy_beg + LS_R * s);
        const char* const rb = rthr + (s & 1) * LS_BUFB;
        f32x2 a[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k][0] = f32x2{0.f, 0.f}; a[k][1] = f32x2{0.f, 0.f}; }
        // two batches of three input rows: 9 reads in flight (the VGPR budget of 3 waves per SIMD).  The accumulators pass through
        // an opaque copy after the first batch so that its FMAs are issued there and not sunk below the second batch's reads.
        if (WL) {
            // tap row by tap row (the same summation order per output: ky, then kx): 3 weight vectors + one slab row at a time
            const char* const wthr = reinterpret_cast<const char*>(&lw[0][0]) + cg * 16;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                f32x4 wr[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) wr[kx] = *reinterpret_cast<const f32x4*>(wthr + (ky * 3 + kx) * LS_PIXB);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 v[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(rb + ((k + ky) * LS_PW + kx) * LS_PIXB);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        a[k][0] = fma2(v[kx].xy, wr[kx].xy, a[k][0]);
                        a[k][1] = fma2(v[kx].zw, wr[kx].zw, a[k][1]);
                    }
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x4 v[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(rb + (r * LS_PW + kx) * LS_PIXB);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int k = r - ky;
                if (k < 0 || k > 3) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    a[k][0] = fma2(v[kx].xy, w[ky * 3 + kx].xy, a[k][0]);
                    a[k][1] = fma2(v[kx].zw, w[ky * 3 + kx].zw, a[k][1]);
                }
            }
            if (r == 2) {
                TSII_PIN_F2(a[0][0]); TSII_PIN_F2(a[0][1]); TSII_PIN_F2(a[1][0]); TSII_PIN_F2(a[1][1]); TSII_PIN_F2(a[2][0]); TSII_PIN_F2(a[2][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float* pls = &lplanes[s & 1][(4 * th) * LS_TW + tx][0];
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (!DXE) bq = *reinterpret_cast<const f32x4*>(cthr + 2 * LS_PIXB);
        float rmk[4] = {0.f, 0.f, 0.f, 0.f};          // K6d: rmask at the thread's pixels (0 for a pixel outside the chunk)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x2 a0 = a[k][0], a1 = a[k][1];
            if (DXE) {
                const float pmk = pls[k * LS_TW * 2];
                if (DWG) rmk[k] = (xok && 4 * th + k < rows_left) ? pmk : 0.f;
                a0 *= pmk; a1 *= pmk;
                if (pmk == 0.f) { a0 = f32x2{0.f, 0.f}; a1 = a0; }
            } else {
                const f32x2 pq = *reinterpret_cast<const f32x2*>(pls + k * LS_TW * 2);
                a0 *= pq.x; a1 *= pq.x;
                a0 += bq.xy; a1 += bq.zw;
                if (pq.y == 0.f) { a0 = f32x2{0.f, 0.f}; a1 = a0; }
            }
            if (H16) { a0 = rnd2(a0); a1 = rnd2(a1); }        // the stored value: statistics / K6c reductions see what memory holds
            a[k][0] = a0; a[k][1] = a1;
            if (xok && 4 * th + k < rows_left) {
                if constexpr (H16) {
                    // two dwords of four bf16: the rounded values are exact in their high halves.  (Scalars first: __builtin_bit_cast
                    // of a vector ELEMENT expression reads element 0 with this clang.)
                    const float e0 = a0.x, e1 = a0.y, e2 = a1.x, e3 = a1.y;
                    const unsigned d0 = (__builtin_bit_cast(unsigned, e0) >> 16) | (__builtin_bit_cast(unsigned, e1) & 0xffff0000u);
                    const unsigned d1 = (__builtin_bit_cast(unsigned, e2) >> 16) | (__builtin_bit_cast(unsigned, e3) & 0xffff0000u);
                    const unsigned long long pk = (unsigned long long)d0 | ((unsigned long long)d1 << 32);
                    __builtin_nontemporal_store(pk, reinterpret_cast<unsigned long long*>(ob + opq((unsigned)(4 * th + k) * orow + ocol)));
                } else
                if (LS_NT_STORE) __builtin_nontemporal_store(cat4(a0, a1), reinterpret_cast<f32x4*>(ob + opq((unsigned)(4 * th + k) * orow + ocol)));
                else *reinterpret_cast<f32x4*>(ob + opq((unsigned)(4 * th + k) * orow + ocol)) = cat4(a0, a1);
                if (FUSED) {
                    if (s == 0 && k == 0) P = cat4(a0, a1);  // a thread with any pixel at all has this one
                    const f32x2 d0 = a0 - P.xy, d1 = a1 - P.zw;
                    va[0] += d0; va[1] += d1;
                    va[2] = fma2(d0, d0, va[2]); va[3] = fma2(d1, d1, va[3]);
                }
            }
        }
        if (BNB) {
            const f32x4 bmu = *reinterpret_cast<const f32x4*>(cthr), bis = *reinterpret_cast<const f32x4*>(cthr + LS_PIXB);
            const f32x4 bga = *reinterpret_cast<const f32x4*>(cthr + 2 * LS_PIXB), bbe = *reinterpret_cast<const f32x4*>(cthr + 3 * LS_PIXB);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 yq = yv[k];
                const f32x2 h0 = (yq.xy - bmu.xy) * bis.xy, h1 = (yq.zw - bmu.zw) * bis.zw;
                const f32x2 z0 = fma2(h0, bga.xy, bbe.xy), z1 = fma2(h1, bga.zw, bbe.zw);
                f32x2 d0 = a[k][0], d1 = a[k][1];
                d0.x *= (z0.x > 0.f && z0.x < bb.hi) ? 1.f : (z0.x > 0.f ? 0.f : bb.neg);
                d0.y *= (z0.y > 0.f && z0.y < bb.hi) ? 1.f : (z0.y > 0.f ? 0.f : bb.neg);
                d1.x *= (z1.x > 0.f && z1.x < bb.hi) ? 1.f : (z1.x > 0.f ? 0.f : bb.neg);
                d1.y *= (z1.y > 0.f && z1.y < bb.hi) ? 1.f : (z1.y > 0.f ? 0.f : bb.neg);
                if (xok && 4 * th + k < rows_left) {
                    va[0] += d0; va[1] += d1;
                    va[2] = fma2(d0, h0, va[2]); va[3] = fma2(d1, h1, va[3]);
                }
                if (DWG) {
                    // the convolution's input at this pixel: act(z) * rmask (exact zero outside the chunk: y was a clamped load)
                    f32x2 e0 = min2(max2(z0, z0 * bb.neg), f32x2{bb.hi, bb.hi}), e1 = min2(max2(z1, z1 * bb.neg), f32x2{bb.hi, bb.hi});
                    e0 *= rmk[k]; e1 *= rmk[k];
                    if (rmk[k] == 0.f) { e0 = f32x2{0.f, 0.f}; e1 = e0; }
                    yv[k] = cat4(e0, e1);
                }
            }
            if (DWG) {
                // second sweep over the step's window: dwa[ky * 3 + kx] += a(q_k) * G(q_k - pad' + (ky, kx)), rows r = k + ky
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    f32x4 v[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(rb + (r * LS_PW + kx) * LS_PIXB);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int k = r - ky;
                        if (k < 0 || k > 3) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            f32x2 lo = dwa[ky * 3 + kx].xy, hi2 = dwa[ky * 3 + kx].zw;
                            lo = fma2(v[kx].xy, yv[k].xy, lo); hi2 = fma2(v[kx].zw, yv[k].zw, hi2);
                            dwa[ky * 3 + kx] = cat4(lo, hi2);
                        }
                    }
                    if (APL && (r & 1)) __builtin_amdgcn_sched_barrier(0);     // K6e: at most two slab rows of the sweep in flight (VGPR budget)
                }
            }
        }
        if (more) { commit((s + 1) & 1); commit_planes(s + 1); }   // into the buffer nobody reads during this step
        if (more2) { fetch(); fetch_planes(s + 2); }
        if (BNB && more) fetch_y(s + 1);
        ob += ob_step;
        lds_barrier();
    };
    for (int s = 0; s + 2 < nsteps; ++s) step(s, true, true);
    if (nsteps > 1) step(nsteps - 2, true, false);
    step(nsteps - 1, false, false);

    if (BNB) {
        float* mrg = lbuf;                               // [256][8]; the loop ended on a barrier: the buffers are free
        float* mt = mrg + t * 8;
        mt[0] = va[0].x; mt[1] = va[0].y; mt[2] = va[1].x; mt[3] = va[1].y;
        mt[4] = va[2].x; mt[5] = va[2].y; mt[6] = va[3].x; mt[7] = va[3].y;
        __syncthreads();
        if (t < 2 * LS_CB) {
            const int which = t / LS_CB, ch = t % LS_CB;
            if ((int)cb * LS_CB + ch < C) {
                float sum = 0.f;
                for (int l = 0; l < 32; ++l) sum += mrg[(l * 8 + ch / 4) * 8 + which * 4 + ch % 4];
                const int64_t prow = (nv * chunks_y + cy) * strips_x + sx;
                bb.part[(prow * 2 + which) * C + (int)cb * LS_CB + ch] = sum;
            }
        }
        if (DWG) {
            // K6d: merge the 32 pixel lanes of every channel, tap by tap: [9][256] float4 through the buffers
            static_assert(9 * 256 * 16 <= 2 * LS_BUFB, "weight-gradient merge buffer fits the LDS buffers");
            __syncthreads();
            f32x4* m4 = reinterpret_cast<f32x4*>(lbuf);
#pragma unroll
            for (int k = 0; k < 9; ++k) m4[k * 256 + t] = dwa[k];
            __syncthreads();
            const int64_t prow = (nv * chunks_y + cy) * strips_x + sx;
            for (int j = t; j < 9 * LS_CB; j += 256) {
                const int k = j / LS_CB, ch = j % LS_CB;
                if ((int)cb * LS_CB + ch >= C) continue;
                const float* col = lbuf + ((k * 256 + ch / 4) * 4 + ch % 4);
                float sum = 0.f;
#pragma unroll 8
                for (int l = 0; l < 32; ++l) sum += col[l * 32];
                const int tap = g.flip ? 8 - k : k;               // the window is the flipped one: tap index of the forward weight
                stats[(prow * 9 + tap) * C + (int)cb * LS_CB + ch] = sum;
            }
        }
    } else if (FUSED) {
        if (stats != nullptr) {
            // merge the 32 pixel lanes of every channel: (count, pivot, s1, s2) per thread through the (now free) buffers,
            // re-based to a common pivot:  s1' = s1 + n dp,  s2' = s2 + 2 dp s1 + n dp^2
            const int rows = oy_end - oy_beg, full = rows / LS_R, tail = rows % LS_R - 4 * th;
            const int cnt = xok ? 4 * full + (tail < 0 ? 0 : (tail > 4 ? 4 : tail)) : 0;
            float* mrg = lbuf;                               // [256][13]
            static_assert(256 * 13 * 4 <= 2 * LS_BUFB, "merge buffer fits the LDS buffers");
            float* mt = mrg + t * 13;
            mt[0] = (float)cnt;
            mt[1] = P.x; mt[2] = P.y; mt[3] = P.z; mt[4] = P.w;
            mt[5] = va[0].x; mt[6] = va[0].y; mt[7] = va[1].x; mt[8] = va[1].y;
            mt[9] = va[2].x; mt[10] = va[2].y; mt[11] = va[3].x; mt[12] = va[3].y;
            __syncthreads();
            if (t < LS_CB && (int)cb * LS_CB + t < C) {
                const int ch = t, mcg = ch / 4, mi = ch % 4;
                float nn = 0.f, pv = 0.f, s1 = 0.f, s2 = 0.f;
                bool have = false;
                {   // common pivot: an interior lane's (lane 17 = rows 4..7, column 1 of the step) when it saw pixels -- the strip's
                    // first pixel is an image-border pixel, the typical outlier of a channel
                    const float* qi = mrg + (17 * 8 + mcg) * 13;
                    if (qi[0] != 0.f) { pv = qi[1 + mi]; have = true; }
                }
                for (int l = 0; l < 32; ++l) {
                    const float* q = mrg + (l * 8 + mcg) * 13;
                    const float n_t = q[0];
                    if (n_t == 0.f) continue;
                    if (!have) { pv = q[1 + mi]; have = true; }
                    const float dp = q[1 + mi] - pv, a1 = q[5 + mi], a2 = q[9 + mi];
                    s1 += fmaf(n_t, dp, a1);
                    s2 += a2 + dp * (2.f * a1 + n_t * dp);
                    nn += n_t;
                }
                const int64_t prow = (nv * chunks_y + cy) * strips_x + sx;      // one partial row per strip chunk
                float* sp = stats + prow * 4 * C + (int)cb * LS_CB + ch;
                sp[0] = nn;
                sp[C] = pv;
                sp[2 * (int64_t)C] = s1;
                sp[3 * (int64_t)C] = s2;
            }
        }
    }
}

// the lean kernel addresses a slab / an output step with 32-bit byte offsets from a wave-uniform base (and forms the activation
// offset with a 24-bit multiply)
#ifndef LS_ENABLE
#define LS_ENABLE 1              // A/B: 0 sends stride-1 / dilation-1 layers back to the round-3 strip kernel
#endif
static inline bool dw_lean_ok(const DtGeom& g) {
    return LS_ENABLE && g.s == 1 && g.d == 1 && g.c < (1 << 24) && (int64_t)(LS_ROWS * (int64_t)g.win + 64) * 4 < (1ll << 24) &&
           (int64_t)(LS_ROWS * (int64_t)g.win + 64) * g.c * 4 < (1ll << 31) &&
           (int64_t)(LS_ROWS * (int64_t)g.wout + 64) * g.c * 4 < (1ll << 31);
}
// dilation by phases (PH): dilations 2 / 4 / 8 whose phase sub-images are at least half a strip wide; every offset of the d = 1
// form times d stays inside the same 24 / 31-bit limits
// Measured (round 6, tools/dw_bench.py cfg3 / dil, profiles/r06e_dw_bench_*.log; phased vs the ring kernels): without mask planes
// dilation 2 forward + K6b 0.321 vs 0.435 ms (64^2 x 768 x 64) and 0.564 vs 0.809 ms (128^2 x 384), dilation 4 0.596 vs 0.681 ms,
// dX + K6c 6-11 % faster -- but dilation 8 on 64^2 maps (8 x 8-pixel phase images: one marching step per block, nothing to
// pipeline) 2.50 vs 2.20 ms, and with mask planes (ImageFill's dilated levels: per-pixel planes read d pixels apart) 10-30 % SLOWER.
// The partial-row layout of the fused forms has to follow from the output grid alone (the row-count entry points see neither the
// mask planes nor the dilation's effect), so the form cannot be chosen per call site: OFF in the stock library (TextSegament's step
// as a whole: 253.4 vs 254.1 ms).  -DLS_PHASED=1 builds it (tools/variants); the CPU suite runs it through an emulator-only switch.
#ifdef TSII_HIP_EMU
static int g_ls_phased = 0;
extern "C" void tsii_emu_set_ls_phased(int v) { g_ls_phased = v; }
#define LS_PHASED g_ls_phased
#else
#ifndef LS_PHASED
#define LS_PHASED 0
#endif
#endif
// (a function of the OUTPUT grid only: the partial-row counts the callers size their buffers with -- tsii_dw_stat_rows /
// tsii_dw_bwd_stat_rows -- do not see the input size; the input is at most 2 d larger per side)
static inline bool dw_phased_dims_ok(int hout, int wout, int c, int s, int d) {
    if (!(LS_PHASED && LS_ENABLE && s == 1 && (d == 2 || d == 4 || d == 8) && c % 4 == 0 && wout >= 8 * d && hout >= 4 * d)) return false;
    const int64_t wmax = (int64_t)wout + 2 * d, hmax = (int64_t)hout + 2 * d;
    return c < (1 << 24) && hmax * wmax * 4 < (1ll << 24) && hmax * wmax * c * 4 < (1ll << 32) &&
           (LS_ROWS * wmax + 64) * c * 4 * d < (1ll << 31);
}
static inline bool dw_lean_phased_ok(const DtGeom& g) {
    return dw_phased_dims_ok(g.hout, g.wout, g.c, g.s, g.d) && g.win <= g.wout + 2 * g.d && g.hin <= g.hout + 2 * g.d && g.pad_h >= 0 && g.pad_w >= 0;
}

}  // namespace tsii
