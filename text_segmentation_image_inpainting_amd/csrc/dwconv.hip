// K2: depth-wise partial convolution (PartialConv with groups == C, same_holes;
// models/partial_convolution.py:49-80 as built at models/MobileNetV2.py:174-176).
//
// HBM-bound streaming stencil on NHWC data: one thread owns 4 consecutive channels (16-byte
// loads, a wavefront covers 256 channels = 1 KiB of one pixel per tap); the x*mask multiply
// (:51), the division by cnt*Cin (:61,71 -- the depth-wise quirk of SURVEY.md F6) and the hole
// zeroing (:72) are fused, so x is read once (taps re-hit L1/L2) and y is written once.
// The valid count comes from K1 (mask.hip) as the `denom`/`keep`/`inv` planes.
#include "tsii_common.h"
#include "dw_lean_api.h"

#include <stdlib.h>

namespace tsii {

struct DwGeom {
    int n, h, w, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo;
};

// weights arrive as [C][T] (reference layout [C,1,kh,kw]); kernels read wT[T][C].
// Grids are (x: pixel groups of a row x channel groups, y: row, z: image): index math is 32-bit.
// Every tap is an UNCONDITIONAL load from a clamped address followed by a select (never `x * 0`: a
// clamped address may hold NaN/Inf) -- data-dependent `continue`s would chain the loads of a pixel
// one after the other; this way a thread keeps PX * taps independent 16-byte loads in flight.
template <bool K3> struct DwPx { static constexpr int value = K3 ? 2 : 4; };  // pixels per thread along x
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int W, bool K3>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rmask,
                                                     const float* __restrict__ wT, const float* __restrict__ bias,
                                                     const float* __restrict__ denom, const float* __restrict__ keep,
                                                     DwGeom g, float* __restrict__ y) {
    constexpr int DW_PX = DwPx<K3>::value;
    const unsigned CG = (unsigned)(g.c / W);
    const unsigned nxg = (unsigned)((g.wo + DW_PX - 1) / DW_PX);
    const unsigned gx = (nxg * CG + 255u) / 256u;               // blocks per output row
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);        // each XCD gets a contiguous band of rows
    const unsigned j = (b % gx) * blockDim.x + threadIdx.x;
    if (j >= nxg * CG) return;
    const unsigned row = b / gx;
    const int ox0 = (int)(j / CG) * DW_PX;
    const int c = (int)(j % CG) * W;
    const int oy = (int)(row % (unsigned)g.ho);
    const int64_t n = row / (unsigned)g.ho;
    float acc[DW_PX][W];
#pragma unroll
    for (int p = 0; p < DW_PX; ++p)
#pragma unroll
        for (int i = 0; i < W; ++i) acc[p][i] = 0.f;
    const int KH = K3 ? 3 : g.kh, KW = K3 ? 3 : g.kw;   // compile-time 3x3: loops unroll, every tap load issues up-front
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int iy = oy * g.sh - g.ph + ky * g.dh;
        const bool yin = (iy >= 0 && iy < g.h);
        const int64_t rowbase = (n * g.h + clampi(iy, 0, g.h - 1)) * g.w;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const VecF<W> wv = vload<W>(wT + (ky * KW + kx) * g.c + c);
#pragma unroll
            for (int p = 0; p < DW_PX; ++p) {
                const int ix = (ox0 + p) * g.sw - g.pw + kx * g.dw;
                const bool inb = yin && ix >= 0 && ix < g.w;
                const int64_t ipix = rowbase + clampi(ix, 0, g.w - 1);
                const float m = rmask != nullptr ? rmask[ipix] : 1.f;
                const VecF<W> xv = vload<W>(x + ipix * g.c + c);
#pragma unroll
                for (int i = 0; i < W; ++i) acc[p][i] = fmaf(inb ? xv.v[i] * m : 0.f, wv.v[i], acc[p][i]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DW_PX; ++p) {
        const int ox = ox0 + p;
        if (ox >= g.wo) break;
        const int64_t pix = (n * g.ho + oy) * g.wo + ox;
        const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;
        const float dn = denom != nullptr ? denom[pix] : 1.f;
        VecF<W> o;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            float v = acc[p][i];
            if (denom != nullptr) v = v / dn;
            if (bias != nullptr) v += bias[c + i];
            o.v[i] = kp ? v : 0.f;
        }
        vstore<W>(y + pix * g.c + c, o);
    }
}

template <int W, bool K3>
__global__ __launch_bounds__(256) void dw_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                        const float* __restrict__ wT, const float* __restrict__ rmask,
                                                        DwGeom g, float* __restrict__ dx) {
    constexpr int DW_PX = DwPx<K3>::value;
    const unsigned CG = (unsigned)(g.c / W);
    const unsigned nxg = (unsigned)((g.w + DW_PX - 1) / DW_PX);
    const unsigned gx = (nxg * CG + 255u) / 256u;
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned j = (b % gx) * blockDim.x + threadIdx.x;
    if (j >= nxg * CG) return;
    const unsigned row = b / gx;
    const int ix0 = (int)(j / CG) * DW_PX;
    const int c = (int)(j % CG) * W;
    const int iy = (int)(row % (unsigned)g.h);
    const int64_t n = row / (unsigned)g.h;
    float acc[DW_PX][W];
#pragma unroll
    for (int p = 0; p < DW_PX; ++p)
#pragma unroll
        for (int i = 0; i < W; ++i) acc[p][i] = 0.f;
    const int KH = K3 ? 3 : g.kh, KW = K3 ? 3 : g.kw;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int ty = iy + g.ph - ky * g.dh;
        const bool vy = ty >= 0 && (ty % g.sh) == 0 && (ty / g.sh) < g.ho;
        const int64_t rowbase = (n * g.ho + clampi(ty / g.sh, 0, g.ho - 1)) * g.wo;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const VecF<W> wv = vload<W>(wT + (ky * KW + kx) * g.c + c);
#pragma unroll
            for (int p = 0; p < DW_PX; ++p) {
                const int tx = ix0 + p + g.pw - kx * g.dw;
                const bool v = vy && tx >= 0 && (tx % g.sw) == 0 && (tx / g.sw) < g.wo;
                const int64_t opix = rowbase + clampi(tx / g.sw, 0, g.wo - 1);
                const float sc = inv != nullptr ? inv[opix] : 1.f;
                const VecF<W> gv = vload<W>(dy + opix * g.c + c);
#pragma unroll
                for (int i = 0; i < W; ++i) acc[p][i] = fmaf(v ? gv.v[i] * sc : 0.f, wv.v[i], acc[p][i]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DW_PX; ++p) {
        const int ix = ix0 + p;
        if (ix >= g.w) break;
        const int64_t pix = (n * g.h + iy) * g.w + ix;
        const float m = rmask != nullptr ? rmask[pix] : 1.f;
        VecF<W> o;
#pragma unroll
        for (int i = 0; i < W; ++i) o.v[i] = (m != 0.f) ? acc[p][i] * m : 0.f;
        vstore<W>(dx + pix * g.c + c, o);
    }
}

// dW partials.  Block = CGB (<= 32) channel groups x L pixel lanes; it walks `rpb` output rows (n,oy),
// lane l taking ox = l, l+L, ...; 9 taps x W channels (+ bias) stay in registers, the L lanes are
// combined through LDS, so each block emits ONE partial row: part[by][(T+1)][C].  Branch-free taps.
static constexpr int DW_TAPS = 9;
template <int W, bool K3>
__global__ __launch_bounds__(256) void dw_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                        const float* __restrict__ keep, const float* __restrict__ x,
                                                        const float* __restrict__ rmask, DwGeom g, int CGB, int L,
                                                        int rpb, float* __restrict__ part) {
    __shared__ float red[256];
    const int CG = g.c / W;
    const int T = g.kh * g.kw;
    const int cgl = threadIdx.x % CGB, lane = threadIdx.x / CGB;
    const unsigned gxb = (unsigned)((CG + CGB - 1) / CGB);
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);        // contiguous row chunks per XCD
    const int bx = (int)(b % gxb), by = (int)(b / gxb);
    const int cg = bx * CGB + cgl;
    const bool active = (lane < L) && (cg < CG);
    const int c = cg * W;
    const int rows_total = g.n * g.ho;
    const int row0 = by * rpb;
    const int row1 = row0 + rpb < rows_total ? row0 + rpb : rows_total;
    float* prow = part + (int64_t)by * (T + 1) * g.c;
    for (int t0 = 0; t0 < T; t0 += DW_TAPS) {
        float acc[DW_TAPS + 1][W];   // last entry: bias gradient
#pragma unroll
        for (int t = 0; t <= DW_TAPS; ++t)
#pragma unroll
            for (int i = 0; i < W; ++i) acc[t][i] = 0.f;
        if (active) {
            for (int row = row0; row < row1; ++row) {
                const int64_t n = row / g.ho;
                const int oy = row % g.ho;
                for (int ox = lane; ox < g.wo; ox += L) {
                    const int64_t pix = (int64_t)row * g.wo + ox;
                    const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;   // hole: no gradient (:72)
                    const float s = inv != nullptr ? inv[pix] : 1.f;
                    const VecF<W> graw = vload<W>(dy + pix * g.c + c);
                    float gv[W];
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        acc[DW_TAPS][i] += kp ? graw.v[i] : 0.f;          // bias is added after the division
                        gv[i] = kp ? graw.v[i] * s : 0.f;
                    }
#pragma unroll
                    for (int t = 0; t < DW_TAPS; ++t) {
                        const int tt = K3 ? t : ((t0 + t < T) ? t0 + t : T - 1);
                        const bool tv = K3 ? true : (t0 + t < T);
                        const int ky = K3 ? t / 3 : tt / g.kw, kx = K3 ? t % 3 : tt % g.kw;
                        const int iy = oy * g.sh - g.ph + ky * g.dh;
                        const int ix = ox * g.sw - g.pw + kx * g.dw;
                        const bool inb = tv && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
                        const int64_t ipix = (n * g.h + clampi(iy, 0, g.h - 1)) * g.w + clampi(ix, 0, g.w - 1);
                        const float m = rmask != nullptr ? rmask[ipix] : 1.f;
                        const VecF<W> xv = vload<W>(x + ipix * g.c + c);
#pragma unroll
                        for (int i = 0; i < W; ++i) acc[t][i] = fmaf(gv[i], inb ? xv.v[i] * m : 0.f, acc[t][i]);
                    }
                }
            }
        }
        // combine the pixel lanes: one value per (tap, channel) at a time through LDS
#pragma unroll
        for (int t = 0; t <= DW_TAPS; ++t) {
            const int tt = (t == DW_TAPS) ? T : t0 + t;
            const bool emit = (t == DW_TAPS) ? (t0 == 0) : (tt < T);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                __syncthreads();
                red[threadIdx.x] = acc[t][i];
                __syncthreads();
                if (emit && active && lane == 0) {
                    float sum = 0.f;
                    for (int l = 0; l < L; ++l) sum += red[l * CGB + cgl];
                    prow[(int64_t)tt * g.c + c + i] = sum;
                }
            }
        }
    }
}

// ---- 3x3 stencils through LDS ------------------------------------------------------------------------------------
// The direct kernels above issue ~20 vector-memory instructions per output pixel and are bound by VMEM issue (measured
// ~1.9 TB/s); the marching-strip kernels below stage the input once in LDS (pre-multiplied by its per-pixel plane: the
// mask for forward, 1/count for dX) and walk the taps out of it with the 9 weights in registers:
//   y[o] = post[o] ? (sum_t w[t] * pre[i_t] * in[i_t]) / denom[o] + bias : 0,   i_t = o*s - pad + t*d
struct DtGeom {
    int n, hin, win, c, s, d, pad_h, pad_w, hout, wout, flip;
};

// InBN (K6b) on the staged input: zero padding stays zero (it pads the activated tensor).
typedef InBN DwBN;

// dX mode of the strip kernel feeding a BatchNorm backward (K6c): the kernel's output IS the gradient w.r.t. the
// normalised activation a = act(gamma*xhat+beta) of the raw tensor `y` (same grid as the output), so the two
// reductions of the BatchNorm backward, sum(dz) and sum(dz*xhat) with dz = out*act'(z), are taken here per strip
// instead of in a separate pass over (out, y).  part: [strip blocks][2][C].
struct DwBnBwd {
    const float* y;
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    float eps, neg, hi;
    float* part;
};

// ---- marching-strip 3x3 stencil, stride 1 (forward, and dX with flipped taps) ------------------------------
// (An 8x16-pixel LDS-tile kernel came first: it re-read a 2d-row halo above and below every tile, 1.4x the input for
// d = 1, and had nothing in flight while it computed -- 1.84 ms where this kernel takes 1.24 ms.)
// A block owns a 16-pixel x 32-channel STRIP and marches down it 8 output rows at a time through a ring of 8 + 2d input rows in LDS: every input row is read once (only the
// 2d-pixel side halo remains, 1.125x), and the next 8 rows (and the per-pixel planes of the next step) are
// already in flight (global -> registers) while the current ones are computed and stored; they are written into
// the ring slots the step has just finished with.  BatchNorm partial sums (K6b) accumulate in registers over the
// whole strip: one reduction / partial row per block.  D (dilation) is a template parameter so that the slab
// geometry (pixel -> row, column) is compile-time arithmetic instead of per-thread index tables.
static constexpr int ST_R = 8, ST_TW = 16, ST_CB = 32;
#ifndef ST_UNROLL
#define ST_UNROLL 2
#endif
static constexpr int ST_UNROLL_K = ST_UNROLL;
#ifndef ST_S2_WAVES
#define ST_S2_WAVES 2     // A/B build knob: waves per SIMD asked of the register allocator for the fused stride-2 forms
#endif   // pixels of a thread processed together (all 4: ~150 VGPRs of LDS data in flight)

// MODE 0: plain; 1: forward with BatchNorm on load and / or statistics partials (K6b; either may be off at run time);
// 2: dX feeding a BatchNorm backward (K6c); 3 (K6d, round 6): MODE 2 that also takes the layer's weight gradient -- the 9 tap values a
// pixel's dX reads from the ring times the layer's input at that pixel (act(bn(y)) * rmask, from the K6c arithmetic) accumulate in
// 9 x 4 registers over the strip chunk, one partial row [9][C] per block in `stats` (see dw_lean.h MODE 3; here for the dilated
// strips, dilation 2 / 4 / 8).  MODE 1 / 2 need ~210 VGPRs = 2 waves per SIMD.  (Tried for MODE 1: a build capped
// at 3 waves per SIMD with the 9 x 4 weights read from LDS per tap and one pixel in flight -- 18-20 spilled registers and
// 17.2 -> 20.4 ms over the depth-wise kernels of a step: the plain-register form stays.)
template <int S, int D, int MODE>
__global__ __launch_bounds__(256, MODE != 0 ? (S == 2 ? ST_S2_WAVES : 2) : 3) void dw_strip_kernel(
    const float* __restrict__ in, const float* __restrict__ pre, const float* __restrict__ wT, const float* __restrict__ bias,
    const float* __restrict__ denom, const float* __restrict__ keep, const float* __restrict__ post_mul, DtGeom g, int chunk_rows,
    unsigned strips_x, unsigned chunks_y, unsigned cblocks, DwBN ib, float* __restrict__ stats, DwBnBwd bb,
    float* __restrict__ out) {
    constexpr bool FUSED = (MODE == 1), BNB = (MODE >= 2), DWG = (MODE == 3);
    static_assert(!DWG || S == 1, "K6d on the ring kernel: stride 1");
    constexpr int R = ST_R / S, TW = ST_TW / S;                 // output rows x columns per step: 8 x 16 at stride 1, 4 x 8 at stride 2
    constexpr int PW = (TW - 1) * S + 2 * D + 1, NR = (R - 1) * S + 2 * D + 1;
    constexpr int NEW = R * S, PRO = NR - NEW;                 // input rows a step brings in / rows the prologue adds first
    constexpr int CGS = ST_CB / 4, LANES = 256 / CGS, NP = R * TW / LANES;
    constexpr int PF = (NEW * PW + LANES - 1) / LANES;         // slab pixels per thread per step
    constexpr int NPX = R * TW;
    constexpr int TYS = LANES / TW;                            // row stride between a thread's pixels
    static_assert(PRO >= 0, "ring rows");
    __shared__ __attribute__((aligned(16))) float ring[NR * PW * ST_CB];
    __shared__ float planes[2][3][NPX];                        // keep / denom / post_mul of a step's pixels, double buffered
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned cb = b % cblocks; b /= cblocks;
    const unsigned sx = b % strips_x; b /= strips_x;
    const unsigned cy = b % chunks_y;
    const int64_t n = b / chunks_y;
    const int cg = threadIdx.x % CGS, lane = threadIdx.x / CGS;
    const int c = (int)cb * ST_CB + cg * 4;
    const bool cok = c < g.c;
    const int oy_beg = (int)cy * chunk_rows;
    const int oy_end = oy_beg + chunk_rows < g.hout ? oy_beg + chunk_rows : g.hout;
    const int ox0 = (int)sx * TW;
    const int iy_base = oy_beg * S - g.pad_h, ix0 = ox0 * S - g.pad_w;   // input row of ring row 0 / input column of slab column 0
    const int nsteps = (oy_end - oy_beg + R - 1) / R;

    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool bn_in = FUSED && ib.sc != nullptr;
    if (bn_in && cok) { isc = *reinterpret_cast<const float4*>(ib.sc + c); ish = *reinterpret_cast<const float4*>(ib.sh + c); }
    float4 w[9];
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4*>(wT + (g.flip ? 8 - t : t) * g.c + c);
        if (bias != nullptr) bq = make_float4(bias[c], bias[c + 1], bias[c + 2], bias[c + 3]);
    }

    const bool bnb = BNB && bb.y != nullptr;
    float4 bmu = make_float4(0.f, 0.f, 0.f, 0.f), bis = bmu, bga = bmu, bbe = bmu;
    if (bnb && cok) {
        bmu = *reinterpret_cast<const float4*>(bb.mean + c);
        const float4 vv = *reinterpret_cast<const float4*>(bb.var + c);
        bis = make_float4(1.0f / sqrtf(vv.x + bb.eps), 1.0f / sqrtf(vv.y + bb.eps), 1.0f / sqrtf(vv.z + bb.eps), 1.0f / sqrtf(vv.w + bb.eps));
        bga = *reinterpret_cast<const float4*>(bb.gamma + c);
        bbe = *reinterpret_cast<const float4*>(bb.beta + c);
    }
    // DEEP (fused forward): a second register set, so the slab of step s + 2 is in flight too (2 waves per SIMD leave the
    // registers for it; one slab of 5 x 16 B per thread in flight does not keep HBM busy at that occupancy)
    constexpr bool DEEP = FUSED;
    float4 pf[PF], pf2[PF];
    float pm[PF], pm2[PF];
    // global -> registers for ring rows [rr0, rr0 + cnt): slab pixel p = lane + 32 i -> (row, px); wave-uniform 64-bit
    // base + 32-bit offsets
    auto fetch = [&](int rr0, int cnt, float4 (&pf)[PF], float (&pm)[PF]) {
        const int iyb = iy_base + rr0;
        const int64_t pixbase = (n * g.hin + iyb) * (int64_t)g.win + ix0;
        const float* __restrict__ src = in + pixbase * g.c + c;
        const float* __restrict__ psrc = pre != nullptr ? pre + pixbase : nullptr;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            const int iy = iyb + row, ix = ix0 + px;
            pf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            pm[i] = 0.f;
            if (row < cnt && cok && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win) {
                const int off = row * g.win + px;
                pf[i] = *reinterpret_cast<const float4*>(src + off * g.c);
                pm[i] = psrc != nullptr ? psrc[off] : 1.f;
            }
        }
    };
    // registers -> ring (BatchNorm + activation of the producer, then the per-pixel plane; out-of-image stays 0)
    auto commit = [&](int rr0, int cnt, const float4 (&pf)[PF], const float (&pm)[PF]) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            if (row >= cnt) continue;
            float4 v = pf[i];
            const float m = pm[i];
            if (bn_in) {
                v.x = bn_act_load(v.x, isc.x, ish.x, ib.neg, ib.hi); v.y = bn_act_load(v.y, isc.y, ish.y, ib.neg, ib.hi);
                v.z = bn_act_load(v.z, isc.z, ish.z, ib.neg, ib.hi); v.w = bn_act_load(v.w, isc.w, ish.w, ib.neg, ib.hi);
            }
            v.x *= m; v.y *= m; v.z *= m; v.w *= m;          // m == 0 outside the image: zero padding of the activated tensor
            *reinterpret_cast<float4*>(ring + (((rr0 + row) % NR) * PW + px) * ST_CB + cg * 4) = v;
        }
    };
    // per-pixel planes of step s (threads 0..127: one pixel each): global -> registers -> LDS, one step ahead
    float pl0 = 1.f, pl1 = 1.f, pl2 = 1.f;
    auto fetch_planes = [&](int s) {
        if (threadIdx.x < NPX) {
            const int oy = oy_beg + R * s + (int)threadIdx.x / TW, ox = ox0 + (int)threadIdx.x % TW;
            const bool ok = oy < oy_end && ox < g.wout;
            const int64_t q = (n * g.hout + (ok ? oy : oy_beg)) * (int64_t)g.wout + (ok ? ox : ox0);
            pl0 = keep != nullptr ? keep[q] : 1.f;
            pl1 = denom != nullptr ? denom[q] : 1.f;
            pl2 = post_mul != nullptr ? post_mul[q] : 1.f;
        }
    };
    auto commit_planes = [&](int s) {
        if (threadIdx.x < NPX) {
            planes[s & 1][0][threadIdx.x] = pl0; planes[s & 1][1][threadIdx.x] = pl1; planes[s & 1][2][threadIdx.x] = pl2;
        }
    };

    fetch_planes(0);
#pragma unroll
    for (int r = 0; r < PRO; r += NEW) {               // halo rows first (more than one round for large dilations)
        fetch(r, PRO - r < NEW ? PRO - r : NEW, pf, pm);
        commit(r, PRO - r < NEW ? PRO - r : NEW, pf, pm);
    }
    fetch(PRO, NEW, pf, pm);
    commit(PRO, NEW, pf, pm);
    commit_planes(0);
    __syncthreads();

    // this thread's output pixels p = lane + 32 k: column tx = lane % TW for every k, row ty = lane / TW + (32 / TW) k
    const int tx = lane % TW, ty0 = lane / TW;
    const bool xok = cok && ox0 + tx < g.wout;
    // BatchNorm partials: thread-local pivot (its first output), merged to a common pivot once at the end
    float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
    float vals[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    float4 dwa[DWG ? 9 : 1];                                 // K6d: weight-gradient accumulators per (flipped) window tap
#pragma unroll
    for (int t = 0; t < (DWG ? 9 : 1); ++t) dwa[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    // one step: fetch slab `fstep` into the (fpf, fpm) set, compute step s, commit slab s + 1 from the (cpf, cpm) set
    // The two barriers of a step guard the LDS ring only and are lds_barrier()s: __syncthreads() also waits for vmcnt(0), i.e. for
    // the slab(s) just requested and for this step's stores -- the prefetch would be over before the compute had begun.
    auto do_step = [&](int s, int fstep, float4 (&fpf)[PF], float (&fpm)[PF], const float4 (&cpf)[PF], const float (&cpm)[PF]) {
        const bool more = s + 1 < nsteps;
        if (fstep < nsteps) fetch(PRO + NEW * fstep, NEW, fpf, fpm);          // in flight during the compute below
        if (more) fetch_planes(s + 1);
        const int oyb = oy_beg + R * s;
        float* __restrict__ out_b = out + ((n * g.hout + oyb) * (int64_t)g.wout + ox0) * g.c + c;
        const float* __restrict__ pls = &planes[s & 1][0][0];
        float4 yv[NP];                                       // K6c: raw BatchNorm input at this thread's output pixels
        if (bnb) {
            const float* __restrict__ y_b = bb.y + ((n * g.hout + oyb) * (int64_t)g.wout + ox0) * g.c + c;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int ty = ty0 + TYS * k;
                yv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xok && oyb + ty < oy_end) yv[k] = *reinterpret_cast<const float4*>(y_b + (ty * g.wout + tx) * g.c);
            }
        }
        // two pixels' LDS reads in flight; K6c: fully unrolled (static yv[k]), one pixel at a time
#pragma unroll (BNB ? NP : ST_UNROLL_K)
        for (int k = 0; k < NP; ++k) {
            if (BNB) __builtin_amdgcn_sched_barrier(0);
            const int ty = ty0 + TYS * k;
            if (!(xok && oyb + ty < oy_end)) continue;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const int pp = ty * TW + tx;
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);      // K6d: the layer's input at this pixel, act(bn(y)) * rmask
            if (DWG) {
                const float pmk = post_mul != nullptr ? pls[2 * NPX + pp] : 1.f;
                if (pmk != 0.f) {
                    const float4 yq = yv[k];
                    const float zx = fmaf((yq.x - bmu.x) * bis.x, bga.x, bbe.x), zy = fmaf((yq.y - bmu.y) * bis.y, bga.y, bbe.y);
                    const float zz = fmaf((yq.z - bmu.z) * bis.z, bga.z, bbe.z), zw = fmaf((yq.w - bmu.w) * bis.w, bga.w, bbe.w);
                    e.x = fminf(fmaxf(zx, zx * bb.neg), bb.hi) * pmk; e.y = fminf(fmaxf(zy, zy * bb.neg), bb.hi) * pmk;
                    e.z = fminf(fmaxf(zz, zz * bb.neg), bb.hi) * pmk; e.w = fminf(fmaxf(zw, zw * bb.neg), bb.hi) * pmk;
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* rp = ring + ((((R * s + ty) * S + ky * D) % NR) * PW + tx * S) * ST_CB + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + kx * D * ST_CB);
                    const float4 ww = w[ky * 3 + kx];
                    a.x = fmaf(v.x, ww.x, a.x); a.y = fmaf(v.y, ww.y, a.y); a.z = fmaf(v.z, ww.z, a.z); a.w = fmaf(v.w, ww.w, a.w);
                    if (DWG) {
                        float4& q = dwa[DWG ? ky * 3 + kx : 0];
                        q.x = fmaf(e.x, v.x, q.x); q.y = fmaf(e.y, v.y, q.y); q.z = fmaf(e.z, v.z, q.z); q.w = fmaf(e.w, v.w, q.w);
                    }
                }
            }
            if (denom != nullptr) { const float rd = 1.0f / pls[NPX + pp]; a.x *= rd; a.y *= rd; a.z *= rd; a.w *= rd; }   // one IEEE division per pixel, then multiplies (<= 1 ulp apart)
            a.x += bq.x; a.y += bq.y; a.z += bq.z; a.w += bq.w;
            if (post_mul != nullptr) {
                const float pmk = pls[2 * NPX + pp];
                a.x = pmk != 0.f ? a.x * pmk : 0.f; a.y = pmk != 0.f ? a.y * pmk : 0.f;
                a.z = pmk != 0.f ? a.z * pmk : 0.f; a.w = pmk != 0.f ? a.w * pmk : 0.f;
            }
            if (keep != nullptr && pls[pp] == 0.f) a = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(out_b + (ty * g.wout + tx) * g.c) = a;
            if (bnb) {
                const float4 yq = yv[k];
                const float hx = (yq.x - bmu.x) * bis.x, hy = (yq.y - bmu.y) * bis.y, hz = (yq.z - bmu.z) * bis.z, hw = (yq.w - bmu.w) * bis.w;
                const float zx = fmaf(hx, bga.x, bbe.x), zy = fmaf(hy, bga.y, bbe.y), zz = fmaf(hz, bga.z, bbe.z), zw = fmaf(hw, bga.w, bbe.w);
                const float dx = a.x * ((zx > 0.f && zx < bb.hi) ? 1.f : (zx > 0.f ? 0.f : bb.neg));
                const float dy = a.y * ((zy > 0.f && zy < bb.hi) ? 1.f : (zy > 0.f ? 0.f : bb.neg));
                const float dz = a.z * ((zz > 0.f && zz < bb.hi) ? 1.f : (zz > 0.f ? 0.f : bb.neg));
                const float dw = a.w * ((zw > 0.f && zw < bb.hi) ? 1.f : (zw > 0.f ? 0.f : bb.neg));
                vals[0] += dx; vals[1] += dy; vals[2] += dz; vals[3] += dw;
                vals[4] = fmaf(dx, hx, vals[4]); vals[5] = fmaf(dy, hy, vals[5]);
                vals[6] = fmaf(dz, hz, vals[6]); vals[7] = fmaf(dw, hw, vals[7]);
            }
            if (FUSED && stats != nullptr) {
                if (cnt == 0) P = a;
                ++cnt;
                const float dx = a.x - P.x, dy = a.y - P.y, dz = a.z - P.z, dw = a.w - P.w;
                vals[0] += dx; vals[1] += dy; vals[2] += dz; vals[3] += dw;
                vals[4] = fmaf(dx, dx, vals[4]); vals[5] = fmaf(dy, dy, vals[5]);
                vals[6] = fmaf(dz, dz, vals[6]); vals[7] = fmaf(dw, dw, vals[7]);
            }
        }
        lds_barrier();                                       // every read of this step's rows is done
        if (more) { commit(PRO + NEW * (s + 1), NEW, cpf, cpm); commit_planes(s + 1); }   // into the slots this step no longer needs
        lds_barrier();
    };
    if constexpr (DEEP) {
        if (nsteps > 1) fetch(PRO + NEW, NEW, pf, pm);
        for (int s = 0; s < nsteps; s += 2) {
            do_step(s, s + 2, pf2, pm2, pf, pm);
            if (s + 1 < nsteps) do_step(s + 1, s + 3, pf, pm, pf2, pm2);
        }
    } else {
        for (int s = 0; s < nsteps; ++s) do_step(s, s + 1, pf, pm, pf, pm);
    }
    if (bnb) {
        float* mrg = ring;                                   // [256][8]
        float* mt = mrg + threadIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) mt[i] = vals[i];
        __syncthreads();
        if (threadIdx.x < 2 * ST_CB) {
            const int which = threadIdx.x / ST_CB, ch = threadIdx.x % ST_CB;
            if ((int)cb * ST_CB + ch < g.c) {
                float sum = 0.f;
                for (int l = 0; l < LANES; ++l) sum += mrg[(l * CGS + ch / 4) * 8 + which * 4 + ch % 4];
                const int64_t prow = (n * chunks_y + cy) * strips_x + sx;
                bb.part[(prow * 2 + which) * g.c + (int)cb * ST_CB + ch] = sum;
            }
        }
        if constexpr (DWG) {
            // K6d: the 32 pixel lanes of every channel, 5 + 4 taps at a time through the ring ([5][256] float4)
            static_assert(5 * 256 * 4 <= NR * PW * ST_CB, "weight-gradient merge buffer fits the ring");
            float4* m4 = reinterpret_cast<float4*>(ring);
            const int64_t prow = (n * chunks_y + cy) * strips_x + sx;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __syncthreads();
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (half * 5 + t < 9) m4[t * 256 + threadIdx.x] = dwa[half * 5 + t < 9 ? half * 5 + t : 0];
                __syncthreads();
                const int nt = half == 0 ? 5 : 4;
                for (int j = threadIdx.x; j < nt * ST_CB; j += 256) {
                    const int t = j / ST_CB, ch = j % ST_CB;
                    if ((int)cb * ST_CB + ch >= g.c) continue;
                    const float* col = ring + (t * 256 + ch / 4) * 4 + ch % 4;
                    float sum = 0.f;
#pragma unroll 8
                    for (int l = 0; l < LANES; ++l) sum += col[l * CGS * 4];
                    const int k = half * 5 + t;
                    stats[(prow * 9 + (g.flip ? 8 - k : k)) * g.c + (int)cb * ST_CB + ch] = sum;
                }
            }
        }
    } else if (FUSED && stats != nullptr) {
        // merge the 32 pixel lanes of every channel: (count, pivot, s1, s2) per thread through the (now free) ring,
        // re-based to the pivot of the first lane that saw a pixel:  s1' = s1 + n dp,  s2' = s2 + 2 dp s1 + n dp^2
        float* mrg = ring;                                   // [256][13]
        static_assert(256 * 13 <= NR * PW * ST_CB, "merge buffer fits the ring");
        float* mt = mrg + threadIdx.x * 13;
        mt[0] = (float)cnt;
        mt[1] = P.x; mt[2] = P.y; mt[3] = P.z; mt[4] = P.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) mt[5 + i] = vals[i];
        __syncthreads();
        if (threadIdx.x < ST_CB && (int)cb * ST_CB + (int)threadIdx.x < g.c) {
            const int ch = threadIdx.x, mcg = ch / 4, mi = ch % 4;
            float nn = 0.f, pv = 0.f, s1 = 0.f, s2 = 0.f;
            bool have = false;
            {   // common pivot: an interior lane's (lane 17 = row 1, column 1 of the step) when it saw pixels -- the strip's
                // first pixel is an image-border pixel, the typical outlier of a channel
                const float* qi = mrg + ((LANES / 2 + 1) * CGS + mcg) * 13;
                if (qi[0] != 0.f) { pv = qi[1 + mi]; have = true; }
            }
            for (int l = 0; l < LANES; ++l) {
                const float* q = mrg + (l * CGS + mcg) * 13;
                const float n_t = q[0];
                if (n_t == 0.f) continue;
                if (!have) { pv = q[1 + mi]; have = true; }
                const float dp = q[1 + mi] - pv, a1 = q[5 + mi], a2 = q[9 + mi];
                s1 += fmaf(n_t, dp, a1);
                s2 += a2 + dp * (2.f * a1 + n_t * dp);
                nn += n_t;
            }
            const int64_t prow = (n * chunks_y + cy) * strips_x + sx;       // one partial row per strip chunk
            float* sp = stats + prow * 4 * g.c + (int)cb * ST_CB + ch;
            sp[0] = nn;
            sp[g.c] = pv;
            sp[2 * (int64_t)g.c] = s1;
            sp[3 * (int64_t)g.c] = s2;
        }
    }
}

// ---- marching-strip dX for 3x3 / stride 2 / pad 1 / dilation 1 --------------------------------------------------
//   dx[iy, ix] = rmask[iy, ix] * sum_{ky,kx : (iy+1-ky), (ix+1-kx) even}  w[ky,kx] * (dy*inv)[(iy+1-ky)/2, (ix+1-kx)/2]
// A block owns a 16-pixel x 32-channel strip of the INPUT grid and marches down it 8 rows at a time; the ring holds the
// 5 x 9 dy*inv pixels a step needs (4 new dy rows per step: dy is a quarter of dx, the kernel is write-bound).  A
// thread's pixels all have the same (row, column) parity -- rows ty0 + 2k, one column -- so its tap set is fixed: two
// candidate taps per axis, the second one zero-weighted where the parity admits a single tap.
// BNB (K6c): dx is the gradient w.r.t. act(bn(y)) of the raw tensor y on the same grid; sum(dz), sum(dz*xhat) per strip
// chunk go to bb.part like in the stride-1 strip kernel.
// DWG (K6d, round 6): the pass also takes the layer's weight gradient.  A thread's (up to) four candidate taps are fixed by its
// parity and their dy*inv values are the ones it has just read for dX; the layer's input at its pixel, act(bn(y)) * rmask, comes
// with the K6c arithmetic: dW[tap] += a(q) * G -- 4 x 4 accumulators, merged per parity class at the end into one partial row
// [9][C] per block (dwpart), instead of dw_strip_dw_kernel<2, 1>'s second pass over (dy, y).
#ifndef DX2_DWG_WAVES
#define DX2_DWG_WAVES 2          // A/B (tools/variants): waves per SIMD the K6d form is compiled for (3: 8 spilled registers; same-box step 60.12 vs 60.36-60.43 ms)
#endif
// APL (K6e, round 6): `dy` is the gradient w.r.t. the ACTIVATION of the BatchNorm that follows the layer (da2), ap.sc that BatchNorm's raw
// input y2 (the layer's output, same [n, ho, wo, c] layout), ap.sh its constants' table coef[6][C], ap.neg / ap.hi its activation: the
// BatchNorm backward is applied while the slab is committed (dw_lean.h MODE 4 has the stride-1 form), outside the tensor the slab stays 0.
template <bool BNB, bool DWG = false, bool APL = false>
__global__ __launch_bounds__(256, DWG ? DX2_DWG_WAVES : 3) void dw_strip_dx2_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                              const float* __restrict__ wT, const float* __restrict__ rmask,
                                                              int n_img, int h, int w_in, int c_all, int ho, int wo, int chunk_rows,
                                                              unsigned strips_x, unsigned chunks_y, unsigned cblocks, DwBnBwd bb,
                                                              float* __restrict__ dx, float* __restrict__ dwpart = nullptr, DwBN ap = DwBN{nullptr, nullptr, 1.f, 0.f}) {
    static_assert(!DWG || BNB, "K6d rides on the K6c form");
    static_assert(!APL || DWG, "K6e rides on the K6d form");
    constexpr int R = 8, TW = 16, NEW = 4, PRO = 1, NR = 5, PW = 9;
    constexpr int CGS = ST_CB / 4, LANES = 256 / CGS, NP = R * TW / LANES, PF = (NEW * PW + LANES - 1) / LANES;
    constexpr int NPX = R * TW;
    __shared__ __attribute__((aligned(16))) float ring[NR * PW * ST_CB];
    __shared__ float planes[2][NPX];                             // rmask of a step's pixels, double buffered
    __shared__ __attribute__((aligned(16))) float dwm[DWG ? 4 * 256 * 4 : 4];   // K6d: the final merge of the weight-gradient accumulators
    float4 dwa[DWG ? 4 : 1];                                     // K6d: candidates AA, AB, BA, BB
#pragma unroll
    for (int i = 0; i < (DWG ? 4 : 1); ++i) dwa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned cb = b % cblocks; b /= cblocks;
    const unsigned sx = b % strips_x; b /= strips_x;
    const unsigned cy = b % chunks_y;
    const int64_t n = b / chunks_y;
    (void)n_img;
    const int cg = threadIdx.x % CGS, lane = threadIdx.x / CGS;
    const int c = (int)cb * ST_CB + cg * 4;
    const bool cok = c < c_all;
    const int iy_beg = (int)cy * chunk_rows;                    // multiple of 8
    const int iy_end = iy_beg + chunk_rows < h ? iy_beg + chunk_rows : h;
    const int ix0 = (int)sx * TW;
    const int oy_base = iy_beg / 2, ox_base = ix0 / 2;          // dy pixel of ring row 0 / slab column 0
    const int nsteps = (iy_end - iy_beg + R - 1) / R;

    float4 pf[PF], pf2[APL ? PF : 1];
    float pm[PF];
    float4 amu = make_float4(0.f, 0.f, 0.f, 0.f), ais = amu, aga = amu, abe = amu, ak1 = amu, ak2 = amu;   // K6e: the folded BatchNorm's constants
    if constexpr (APL) {
        if (cok) {
            const float* cf = ap.sh + c;
            amu = *reinterpret_cast<const float4*>(cf); ais = *reinterpret_cast<const float4*>(cf + c_all);
            aga = *reinterpret_cast<const float4*>(cf + 2 * (int64_t)c_all); abe = *reinterpret_cast<const float4*>(cf + 3 * (int64_t)c_all);
            ak1 = *reinterpret_cast<const float4*>(cf + 4 * (int64_t)c_all); ak2 = *reinterpret_cast<const float4*>(cf + 5 * (int64_t)c_all);
        }
    }
    auto fetch = [&](int rr0, int cnt) {
        const int oyb = oy_base + rr0;
        const int64_t pixbase = (n * ho + oyb) * (int64_t)wo + ox_base;
        const float* __restrict__ src = dy + pixbase * c_all + c;
        const float* __restrict__ src2 = APL ? ap.sc + pixbase * c_all + c : nullptr;
        const float* __restrict__ psrc = inv != nullptr ? inv + pixbase : nullptr;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            const int oy = oyb + row, ox = ox_base + px;
            pf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (APL) pf2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            pm[i] = 0.f;
            if (row < cnt && cok && oy < ho && ox < wo) {
                const int off = row * wo + px;
                pf[i] = *reinterpret_cast<const float4*>(src + off * c_all);
                if constexpr (APL) pf2[i] = *reinterpret_cast<const float4*>(src2 + off * c_all);
                pm[i] = psrc != nullptr ? psrc[off] : 1.f;
            }
        }
    };
    // K6e: one element of the folded BatchNorm's backward, the stand-alone pass's arithmetic in its order (bn_bwd_apply_kernel)
    auto bn2 = [&](float da, float y2, float mu, float is, float ga, float be, float k1, float k2) {
        const float xh = (y2 - mu) * is;
        const float z = xh * ga + be;
        float dz = da * ((z > 0.f && z < ap.hi) ? 1.f : (z > 0.f ? 0.f : ap.neg));
        dz = dz - k1 - xh * k2;
        return dz * ga * is;
    };
    auto commit = [&](int rr0, int cnt) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            if (row >= cnt) continue;
            float4 v = pf[i];
            const float m = pm[i];
            if constexpr (APL) {
                const float4 y2 = pf2[APL ? i : 0];
                v.x = bn2(v.x, y2.x, amu.x, ais.x, aga.x, abe.x, ak1.x, ak2.x); v.y = bn2(v.y, y2.y, amu.y, ais.y, aga.y, abe.y, ak1.y, ak2.y);
                v.z = bn2(v.z, y2.z, amu.z, ais.z, aga.z, abe.z, ak1.z, ak2.z); v.w = bn2(v.w, y2.w, amu.w, ais.w, aga.w, abe.w, ak1.w, ak2.w);
            }
            v.x *= m; v.y *= m; v.z *= m; v.w *= m;          // m == 0 outside the tensor: the slab stays 0 there (the transform of zeros is not)
            *reinterpret_cast<float4*>(ring + (((rr0 + row) % NR) * PW + px) * ST_CB + cg * 4) = v;
        }
    };
    float pl0 = 1.f;
    auto fetch_planes = [&](int s) {
        if (threadIdx.x < NPX) {
            const int iy = iy_beg + R * s + (int)threadIdx.x / TW, ix = ix0 + (int)threadIdx.x % TW;
            const bool ok = iy < iy_end && ix < w_in;
            pl0 = (rmask != nullptr && ok) ? rmask[(n * h + iy) * (int64_t)w_in + ix] : 1.f;
        }
    };
    auto commit_planes = [&](int s) {
        if (threadIdx.x < NPX) planes[s & 1][threadIdx.x] = pl0;
    };

    // this thread's pixels: column tx, rows ty0 + 2k -> fixed parity; candidate taps (A, B) per axis
    const int tx = lane % TW, ty0 = lane / TW;
    const bool xok = cok && ix0 + tx < w_in;
    const bool ey = ((ty0 + 1) & 1) == 0, ex = ((tx + 1) & 1) == 0;   // (i + pad) even: taps 0 and 2, else tap 1 only
    const int kyA = ey ? 0 : 1, kyB = ey ? 2 : 1, kxA = ex ? 0 : 1, kxB = ex ? 2 : 1;
    float4 wAA = make_float4(0.f, 0.f, 0.f, 0.f), wAB = wAA, wBA = wAA, wBB = wAA;
    if (cok) {
        wAA = *reinterpret_cast<const float4*>(wT + (kyA * 3 + kxA) * c_all + c);
        if (ex) wAB = *reinterpret_cast<const float4*>(wT + (kyA * 3 + kxB) * c_all + c);
        if (ey) wBA = *reinterpret_cast<const float4*>(wT + (kyB * 3 + kxA) * c_all + c);
        if (ey && ex) wBB = *reinterpret_cast<const float4*>(wT + (kyB * 3 + kxB) * c_all + c);
    }
    const int cxA = (tx + 1 - kxA) / 2, cxB = (tx + 1 - kxB) / 2;     // slab columns of the two candidates
    float4 bmu = make_float4(0.f, 0.f, 0.f, 0.f), bis = bmu, bga = bmu, bbe = bmu;
    float vals[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (BNB) {
        if (cok) {
            bmu = *reinterpret_cast<const float4*>(bb.mean + c);
            const float4 vv = *reinterpret_cast<const float4*>(bb.var + c);
            bis = make_float4(1.0f / sqrtf(vv.x + bb.eps), 1.0f / sqrtf(vv.y + bb.eps), 1.0f / sqrtf(vv.z + bb.eps), 1.0f / sqrtf(vv.w + bb.eps));
            bga = *reinterpret_cast<const float4*>(bb.gamma + c);
            bbe = *reinterpret_cast<const float4*>(bb.beta + c);
        }
    }

    fetch_planes(0);
    fetch(0, PRO);
    commit(0, PRO);
    fetch(PRO, NEW);
    commit(PRO, NEW);
    commit_planes(0);
    __syncthreads();

#ifndef DX2_Y_AHEAD
#define DX2_Y_AHEAD 1            // A/B (tools/variants): the K6c input of step s + 1 requested during step s (0: at the top of its own step)
#endif
    // K6c: raw BatchNorm input at this thread's pixels, requested one step ahead like the slab (round 6: requested at the top of its own
    // step it was a memory round trip per step that only the step's 4 pixels of arithmetic could hide)
    float4 yn[BNB ? NP : 1];
    auto fetch_y = [&](int s) {
        if constexpr (BNB) {
            const int iyb = iy_beg + R * s;
            const float* __restrict__ y_b = bb.y + ((n * h + iyb) * (int64_t)w_in + ix0) * c_all + c;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int ty = ty0 + 2 * k;
                yn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xok && iyb + ty < iy_end) yn[k] = *reinterpret_cast<const float4*>(y_b + (ty * w_in + tx) * c_all);
            }
        }
    };
    if (DX2_Y_AHEAD) fetch_y(0);
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        if (more) { fetch(PRO + NEW * (s + 1), NEW); fetch_planes(s + 1); }
        const int iyb = iy_beg + R * s;
        float* __restrict__ out_b = dx + ((n * h + iyb) * (int64_t)w_in + ix0) * c_all + c;
        const float* __restrict__ pls = &planes[s & 1][0];
        float4 yv[BNB ? NP : 1];
        if constexpr (BNB) {
            if (!DX2_Y_AHEAD) fetch_y(s);
#pragma unroll
            for (int k = 0; k < NP; ++k) yv[k] = yn[k];
            if (DX2_Y_AHEAD && more) fetch_y(s + 1);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int ty = ty0 + 2 * k;
            if (!(xok && iyb + ty < iy_end)) continue;
            const int rA = (4 * s + (ty + 1 - kyA) / 2) % NR, rB = (4 * s + (ty + 1 - kyB) / 2) % NR;
            const float4 vAA = *reinterpret_cast<const float4*>(ring + (rA * PW + cxA) * ST_CB + cg * 4);
            const float4 vAB = *reinterpret_cast<const float4*>(ring + (rA * PW + cxB) * ST_CB + cg * 4);
            const float4 vBA = *reinterpret_cast<const float4*>(ring + (rB * PW + cxA) * ST_CB + cg * 4);
            const float4 vBB = *reinterpret_cast<const float4*>(ring + (rB * PW + cxB) * ST_CB + cg * 4);
            float4 a;
            a.x = fmaf(vAA.x, wAA.x, fmaf(vAB.x, wAB.x, fmaf(vBA.x, wBA.x, vBB.x * wBB.x)));
            a.y = fmaf(vAA.y, wAA.y, fmaf(vAB.y, wAB.y, fmaf(vBA.y, wBA.y, vBB.y * wBB.y)));
            a.z = fmaf(vAA.z, wAA.z, fmaf(vAB.z, wAB.z, fmaf(vBA.z, wBA.z, vBB.z * wBB.z)));
            a.w = fmaf(vAA.w, wAA.w, fmaf(vAB.w, wAB.w, fmaf(vBA.w, wBA.w, vBB.w * wBB.w)));
            const float pmk = pls[ty * TW + tx];
            a.x = pmk != 0.f ? a.x * pmk : 0.f; a.y = pmk != 0.f ? a.y * pmk : 0.f;
            a.z = pmk != 0.f ? a.z * pmk : 0.f; a.w = pmk != 0.f ? a.w * pmk : 0.f;
            *reinterpret_cast<float4*>(out_b + (ty * w_in + tx) * c_all) = a;
            if constexpr (BNB) {
                const float4 yq = yv[k];
                const float hx = (yq.x - bmu.x) * bis.x, hy = (yq.y - bmu.y) * bis.y, hz = (yq.z - bmu.z) * bis.z, hw = (yq.w - bmu.w) * bis.w;
                const float zx = fmaf(hx, bga.x, bbe.x), zy = fmaf(hy, bga.y, bbe.y), zz = fmaf(hz, bga.z, bbe.z), zw = fmaf(hw, bga.w, bbe.w);
                const float gx = a.x * ((zx > 0.f && zx < bb.hi) ? 1.f : (zx > 0.f ? 0.f : bb.neg));
                const float gy = a.y * ((zy > 0.f && zy < bb.hi) ? 1.f : (zy > 0.f ? 0.f : bb.neg));
                const float gz = a.z * ((zz > 0.f && zz < bb.hi) ? 1.f : (zz > 0.f ? 0.f : bb.neg));
                const float gw = a.w * ((zw > 0.f && zw < bb.hi) ? 1.f : (zw > 0.f ? 0.f : bb.neg));
                vals[0] += gx; vals[1] += gy; vals[2] += gz; vals[3] += gw;
                vals[4] = fmaf(gx, hx, vals[4]); vals[5] = fmaf(gy, hy, vals[5]);
                vals[6] = fmaf(gz, hz, vals[6]); vals[7] = fmaf(gw, hw, vals[7]);
                if constexpr (DWG) {
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);      // the layer's input at this pixel: act(z) * rmask
                    if (pmk != 0.f) {
                        e.x = fminf(fmaxf(zx, zx * bb.neg), bb.hi) * pmk; e.y = fminf(fmaxf(zy, zy * bb.neg), bb.hi) * pmk;
                        e.z = fminf(fmaxf(zz, zz * bb.neg), bb.hi) * pmk; e.w = fminf(fmaxf(zw, zw * bb.neg), bb.hi) * pmk;
                    }
                    dwa[0].x = fmaf(e.x, vAA.x, dwa[0].x); dwa[0].y = fmaf(e.y, vAA.y, dwa[0].y); dwa[0].z = fmaf(e.z, vAA.z, dwa[0].z); dwa[0].w = fmaf(e.w, vAA.w, dwa[0].w);
                    dwa[1].x = fmaf(e.x, vAB.x, dwa[1].x); dwa[1].y = fmaf(e.y, vAB.y, dwa[1].y); dwa[1].z = fmaf(e.z, vAB.z, dwa[1].z); dwa[1].w = fmaf(e.w, vAB.w, dwa[1].w);
                    dwa[2].x = fmaf(e.x, vBA.x, dwa[2].x); dwa[2].y = fmaf(e.y, vBA.y, dwa[2].y); dwa[2].z = fmaf(e.z, vBA.z, dwa[2].z); dwa[2].w = fmaf(e.w, vBA.w, dwa[2].w);
                    dwa[3].x = fmaf(e.x, vBB.x, dwa[3].x); dwa[3].y = fmaf(e.y, vBB.y, dwa[3].y); dwa[3].z = fmaf(e.z, vBB.z, dwa[3].z); dwa[3].w = fmaf(e.w, vBB.w, dwa[3].w);
                }
            }
        }
        lds_barrier();
        if (more) { commit(PRO + NEW * (s + 1), NEW); commit_planes(s + 1); }
        lds_barrier();
    }
    if constexpr (BNB) {
        // lanes of a wave with the same channel group sit 8 threads apart: butterfly over them, then 4 wave rows through LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = vals[i];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            vals[i] = v;
        }
        float* mrg = ring;                                   // [4 waves][CGS][8]; the loop ended on a barrier: the ring is free
        if ((threadIdx.x & 63) < CGS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mrg[((threadIdx.x >> 6) * CGS + cg) * 8 + i] = vals[i];
        }
        __syncthreads();
        if (threadIdx.x < 2 * ST_CB) {
            const int which = threadIdx.x / ST_CB, ch = threadIdx.x % ST_CB;
            if ((int)cb * ST_CB + ch < c_all) {
                float sum = 0.f;
                for (int wv = 0; wv < 4; ++wv) sum += mrg[(wv * CGS + ch / 4) * 8 + which * 4 + ch % 4];
                const int64_t prow = (n * chunks_y + cy) * strips_x + sx;
                bb.part[(prow * 2 + which) * c_all + (int)cb * ST_CB + ch] = sum;
            }
        }
    }
    if constexpr (DWG) {
        // tap (ky, kx) collects the candidate slot (ky == 2 ? B : A, kx == 2 ? B : A) of the 8 pixel lanes whose parity admits it:
        // rows ty0 = (ky == 1 ? 0 : 1), columns tx even for kx == 1, odd otherwise (a thread without the second candidate on an axis
        // holds a duplicate of the first in that slot: never read)
        float4* m4 = reinterpret_cast<float4*>(dwm);
#pragma unroll
        for (int i = 0; i < 4; ++i) m4[i * 256 + threadIdx.x] = dwa[i];
        __syncthreads();
        const int64_t prow = (n * chunks_y + cy) * strips_x + sx;
        for (int j = threadIdx.x; j < 9 * ST_CB; j += 256) {
            const int tap = j / ST_CB, ch = j % ST_CB;
            if ((int)cb * ST_CB + ch >= c_all) continue;
            const int ky = tap / 3, kx = tap % 3;
            const int slot = (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0);
            const int ty_sel = ky == 1 ? 0 : 1, tx_par = kx == 1 ? 0 : 1;
            float sum = 0.f;
#pragma unroll
            for (int l = 0; l < TW / 2; ++l) {
                const int ln = ty_sel * TW + 2 * l + tx_par;          // pixel lane
                sum += dwm[((slot * 256 + ln * CGS + ch / 4) * 4) + ch % 4];
            }
            dwpart[(prow * 9 + tap) * c_all + (int)cb * ST_CB + ch] = sum;
        }
    }
}

}  // namespace tsii
#include "dw_lean.h"
#include "dw_lean_s2.h"
#include "dw_small.h"
#include "dw_rows.h"
namespace tsii {

struct StripPlan {
    bool ok;
    int chunk_rows;
    unsigned strips_x, chunks_y, cblocks;
};
#ifdef TSII_HIP_EMU
// TEST-ONLY (emulator build, tests/emu): the block target of plan_strip, so that small test tensors get chunks of several steps
static int g_strip_target = 1536;
extern "C" void tsii_emu_set_strip_target(int v) { g_strip_target = v > 0 ? v : 1536; }
#else
#ifndef LS_TARGET
#define LS_TARGET 1536
#endif
static constexpr int g_strip_target = LS_TARGET;
#endif
static StripPlan plan_strip(int n, int hout, int wout, int c, int s, int d) {
    StripPlan p;
    p.ok = ((s == 1 && (d == 1 || d == 2 || d == 4 || d == 8)) || (s == 2 && d == 1)) && c % 4 == 0;
    p.strips_x = cdiv(wout, ST_TW / (s == 2 ? 2 : 1));
    p.cblocks = cdiv(c, ST_CB);
    const int64_t per_chunk = (int64_t)p.strips_x * p.cblocks * n;
    int64_t want = cdiv64(g_strip_target, per_chunk);        // ~6 blocks per CU
    const int max_chunks = cdiv(hout, ST_R);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    p.chunk_rows = cdiv(cdiv(hout, (int)want), ST_R) * ST_R;
    p.chunks_y = cdiv(hout, p.chunk_rows);
    if (per_chunk * p.chunks_y >= (1ll << 31)) p.ok = false;
    return p;
}

// partial rows of the fused forms: ONE definition for the strip kernels, the small-map kernel (which writes the same layout) and the
// row counts the callers size their buffers with (tsii_dw_stat_rows / tsii_dw_bwd_stat_rows)
static inline unsigned strip_rows_per_image(const StripPlan& sp) { return sp.ok ? sp.chunks_y * sp.strips_x : 0u; }

// The plan of a stride-1 layer's forward / dX strips over the OUTPUT grid [hout, wout]: dilations 2 / 4 / 8 run the lean kernel per
// phase (dw_lean.h, PH) -- d^2 sub-images of every d-th row and column per image, planned as n d^2 images of ceil(h / d) x ceil(w / d)
// pixels; `phases` = d^2 then, else 1.  rows per (real) image = phases * chunks_y * strips_x.
struct FusedPlan {
    StripPlan sp;
    unsigned phases;
    bool rows_only;      // no strip kernel for this dilation: the plan only sizes the row-phase kernel's partial rows (dw_rows.h)
};
static FusedPlan plan_fwd_strips(int n, int hout, int wout, int c, int s, int d) {
    FusedPlan fp;
    fp.rows_only = false;
    if (dw_phased_dims_ok(hout, wout, c, s, d)) {
        fp.phases = (unsigned)(d * d);
        if ((int64_t)n * d * d < (1ll << 24)) {
            fp.sp = plan_strip(n * d * d, cdiv(hout, d), cdiv(wout, d), c, 1, 1);
            return fp;
        }
    }
    fp.phases = 1u;
    fp.sp = plan_strip(n, hout, wout, c, s, d);
    if (!fp.sp.ok && dw_rows_dims_ok(hout, wout, c, s, d)) {
        // a dilation without strip form (16, 17, 29 ...) on a map the row-phase kernel takes (dw_rows.h): one partial row per image
        fp.sp.ok = true; fp.sp.chunk_rows = hout; fp.sp.strips_x = 1; fp.sp.chunks_y = 1; fp.sp.cblocks = (unsigned)cdiv(c, ST_CB);
        fp.rows_only = true;
    }
    return fp;
}
static inline unsigned fused_rows_per_image(const FusedPlan& fp) { return fp.phases * strip_rows_per_image(fp.sp); }

// -> 0 launched, 1 not applicable (caller falls back to the direct kernel), <0 error
static const DwBN kNoDwBN = {nullptr, nullptr, 1.f, 0.f};
static const DwBnBwd kNoBnBwd = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr};

// geometries with the K6b / K6c strip variants (stride 2: forward only here; its dX has its own kernel)
static bool dw_fused_ok(int s, int d) { return (s == 2 && d == 1) || (s == 1 && (d == 1 || d == 2 || d == 4 || d == 8)); }

// K6d: the geometry (of the dX grid: g.hout x g.wout = the layer's input) the lean kernel takes before any other strip path does
#ifndef ST_DWG
#define ST_DWG 1                 // A/B (tools/variants): 0 = no K6d on the dilated ring kernels
#endif
#ifndef SM_DWG
#define SM_DWG 1                 // A/B (tools/variants): 0 = no K6d on the small-map kernel
#endif
static bool dw_dxdw_geom_ok(const DtGeom& g) {
    if (g.s != 1 || g.flip != 1 || g.pad_h < 0 || g.pad_w < 0 || g.c % 4 != 0 || dw_rows_ok(g)) return false;
    if (dw_small_ok(g)) return SM_DWG && dw_fused_ok(g.s, g.d);          // the small-map kernel (dw_small_kernel<3>) where it has the K6c form
    if (g.d == 1) return dw_lean_ok(g);
    return ST_DWG && (g.d == 2 || g.d == 4 || g.d == 8);          // the ring kernels (dw_strip_kernel<1, D, 3>)
}

static int try_launch_dw_strip(const float* in, const float* pre, const float* wT, const float* bias, const float* denom,
                               const float* keep, const float* post_mul, DtGeom g, float* out, hipStream_t st,
                               DwBN ib = kNoDwBN, float* stats = nullptr, DwBnBwd bb = kNoBnBwd, float* dwpart = nullptr, bool fold = false) {
    if (g.c % 4 != 0 || !aligned16(in) || !aligned16(out) || !aligned16(wT)) return 1;
    if (ib.sc != nullptr && (!aligned16(ib.sc) || !aligned16(ib.sh))) return 1;
    const FusedPlan fp = plan_fwd_strips(g.n, g.hout, g.wout, g.c, g.s, g.d);
    const StripPlan sp = fp.sp;
    // K6d (dX + K6c + the weight gradient in one pass, dwpart = [partial rows][9][C]) exists on the lean kernel only
    // (K6e, fold: `in` is the gradient w.r.t. the activation of the BatchNorm that FOLLOWS the layer; ib carries that BatchNorm's raw
    // input, its constants' table and its activation -- dw_lean.h MODE 4)
    if (dwpart != nullptr && !(dw_dxdw_geom_ok(g) && sp.ok && fp.phases == 1 && !fp.rows_only && bb.y != nullptr && stats == nullptr &&
                               (ib.sc == nullptr) == !fold && denom == nullptr && keep == nullptr && bias == nullptr)) return 1;
    if (fold && dwpart == nullptr) return 1;
    const bool fused_any = ib.sc != nullptr || stats != nullptr || bb.y != nullptr;      // (fold: bb.y is set anyway)
    // large dilation on mid-sized maps, no mask planes: one row phase of a channel block in LDS (dw_rows.h)
    if (dw_rows_ok(g) && pre == nullptr && denom == nullptr && keep == nullptr && post_mul == nullptr && sp.ok &&
        (int64_t)g.n * cdiv(g.c, DR_CB) < (1ll << 31)) {
        const unsigned rcb = (unsigned)cdiv(g.c, DR_CB), rpi = fused_rows_per_image(fp);
        const dim3 rgrid((unsigned)g.n * rcb);
        if (bb.y != nullptr) hipLaunchKernelGGL((dw_rows_kernel<2>), rgrid, dim3(DR_THREADS), 0, st, in, wT, bias, g, rcb, rpi, ib, stats, bb, out);
        else if (fused_any) hipLaunchKernelGGL((dw_rows_kernel<1>), rgrid, dim3(DR_THREADS), 0, st, in, wT, bias, g, rcb, rpi, ib, stats, bb, out);
        else hipLaunchKernelGGL((dw_rows_kernel<0>), rgrid, dim3(DR_THREADS), 0, st, in, wT, bias, g, rcb, rpi, ib, stats, bb, out);
        return check_launch("dw_rows");
    }
    if (fp.rows_only) {          // mask planes / another padding at a dilation without strip kernels: the direct kernels, unfused only
        TSII_REQUIRE(!fused_any, "depth-wise strips: dilation %d has a fused BatchNorm form on the row-phase kernel only (no mask planes, padding = dilation)", g.d);
        return 1;
    }
    // small maps: the whole map of a channel block in LDS (any dilation; the fused forms where the strip plan defines the partial rows)
    if (dw_small_ok(g) && !(post_mul != nullptr && (denom != nullptr || keep != nullptr || bias != nullptr)) && (!fused_any || (sp.ok && dw_fused_ok(g.s, g.d))) &&
        !(fused_any && bb.y == nullptr && post_mul != nullptr)) {
        const bool dxe = denom == nullptr && keep == nullptr && bias == nullptr;
        if (bb.y == nullptr || dxe) {
            const unsigned scb = (unsigned)cdiv(g.c, SM_CB), rpi = sp.ok ? fused_rows_per_image(fp) : 1u;
            const dim3 sgrid((unsigned)g.n * scb);
#define TSII_DW_SMALL(MODE, DXE) do { \
            if (pre != nullptr) hipLaunchKernelGGL((dw_small_kernel<MODE, DXE, true>), sgrid, dim3(SM_THREADS), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, scb, rpi, ib, stats, bb, out); \
            else hipLaunchKernelGGL((dw_small_kernel<MODE, DXE, false>), sgrid, dim3(SM_THREADS), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, scb, rpi, ib, stats, bb, out); } while (0)
            if (bb.y != nullptr && dwpart != nullptr) {
                if (pre != nullptr) hipLaunchKernelGGL((dw_small_kernel<3, true, true>), sgrid, dim3(SM_THREADS), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, scb, rpi, ib, dwpart, bb, out);
                else hipLaunchKernelGGL((dw_small_kernel<3, true, false>), sgrid, dim3(SM_THREADS), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, scb, rpi, ib, dwpart, bb, out);
            }
            else if (bb.y != nullptr) TSII_DW_SMALL(2, true);
            else if (fused_any) TSII_DW_SMALL(1, false);
            else if (dxe) TSII_DW_SMALL(0, true);
            else TSII_DW_SMALL(0, false);
#undef TSII_DW_SMALL
            return check_launch("dw_small");
        }
    }
    if (!sp.ok) return 1;
    const int64_t nblk = (int64_t)sp.strips_x * sp.chunks_y * sp.cblocks * g.n * fp.phases;
    if (nblk >= (1ll << 31)) return fp.phases > 1 ? -1 : 1;
    const dim3 grid((unsigned)nblk);
    const bool fused = ib.sc != nullptr || stats != nullptr || bb.y != nullptr;
    if (!dw_fused_ok(g.s, g.d)) {
        // (a dilation whose only fused form is the row-phase kernel above, called with mask planes or another padding)
        TSII_REQUIRE(!fused, "depth-wise strips: dilation %d has a fused BatchNorm form on the row-phase kernel only (no mask planes, padding = dilation)", g.d);
        return 1;
    }
    if (bb.y != nullptr && g.s != 1) return 1;
    if (fp.phases > 1) {
        // dilation by phases: the partial rows of the fused forms follow THIS plan (tsii_dw_stat_rows / tsii_dw_bwd_stat_rows), so a
        // geometry the phased kernel cannot take is an error for them, not a fall-through to a kernel with another row layout
        const bool takes = dw_lean_phased_ok(g) && !(post_mul != nullptr && (denom != nullptr || keep != nullptr || bias != nullptr));
        const bool dxe = denom == nullptr && keep == nullptr && bias == nullptr;
        if (!takes || (bb.y != nullptr && !dxe) || (fused && bb.y == nullptr && post_mul != nullptr)) {
            TSII_REQUIRE(!fused, "depth-wise strips: this dilated geometry has no fused BatchNorm form (input more than 2 d larger than the output per side?)");
            return 1;
        }
#define TSII_DW_LEAN_PH(MODE, DXE) do { \
            if (pre != nullptr) hipLaunchKernelGGL((dw_lean_kernel<MODE, DXE, true, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out); \
            else hipLaunchKernelGGL((dw_lean_kernel<MODE, DXE, false, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                    sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out); } while (0)
        if (bb.y != nullptr) TSII_DW_LEAN_PH(2, true);
        else if (fused) TSII_DW_LEAN_PH(1, false);
        else if (dxe) TSII_DW_LEAN_PH(0, true);
        else TSII_DW_LEAN_PH(0, false);
#undef TSII_DW_LEAN_PH
        return check_launch("dw_lean_phased");
    }
    if (dw_lean_ok(g) && !(post_mul != nullptr && (denom != nullptr || keep != nullptr || bias != nullptr))) {
        const bool dxe = denom == nullptr && keep == nullptr && bias == nullptr;
#define TSII_DW_LEAN(MODE, DXE) do { \
            if (pre != nullptr) hipLaunchKernelGGL((dw_lean_kernel<MODE, DXE, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out); \
            else hipLaunchKernelGGL((dw_lean_kernel<MODE, DXE, false>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                    sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out); } while (0)
        if (bb.y != nullptr) {
            if (!dxe) return 1;
            if (dwpart != nullptr && fold) {
                if (pre != nullptr) hipLaunchKernelGGL((dw_lean_kernel<4, true, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                                       sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, dwpart, bb, out);
                else hipLaunchKernelGGL((dw_lean_kernel<4, true, false>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                        sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, dwpart, bb, out);
                return check_launch("dw_lean (BatchNorm backward on load + dX + dW)");
            }
            if (dwpart != nullptr) {
                if (pre != nullptr) hipLaunchKernelGGL((dw_lean_kernel<3, true, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                                       sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, dwpart, bb, out);
                else hipLaunchKernelGGL((dw_lean_kernel<3, true, false>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                        sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, dwpart, bb, out);
                return check_launch("dw_lean (dX + dW)");
            }
            TSII_DW_LEAN(2, true);
        } else if (fused) {
            if (post_mul != nullptr) return 1;
            TSII_DW_LEAN(1, false);
        } else if (dxe) TSII_DW_LEAN(0, true);
        else TSII_DW_LEAN(0, false);
#undef TSII_DW_LEAN
        return check_launch("dw_lean");
    }
    if (dw_lean_s2_ok(g) && post_mul == nullptr && bb.y == nullptr) {
#define TSII_DW_LEAN_S2(MODE) do { \
            if (pre != nullptr) hipLaunchKernelGGL((dw_lean_s2_kernel<MODE, true>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, g, \
                                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, out); \
            else hipLaunchKernelGGL((dw_lean_s2_kernel<MODE, false>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, g, \
                                    sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, out); } while (0)
        if (fused) TSII_DW_LEAN_S2(1);
        else TSII_DW_LEAN_S2(0);
#undef TSII_DW_LEAN_S2
        return check_launch("dw_lean_s2");
    }
#define TSII_DW_STRIP(S, D, MODE) hipLaunchKernelGGL((dw_strip_kernel<S, D, MODE>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                                     sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out)
#define TSII_DW_STRIP_D(D) do { if (bb.y != nullptr && dwpart != nullptr) hipLaunchKernelGGL((dw_strip_kernel<1, D, 3>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g, \
                                                                                          sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, dwpart, bb, out); \
                                else if (bb.y != nullptr) TSII_DW_STRIP(1, D, 2); else if (fused) TSII_DW_STRIP(1, D, 1); else TSII_DW_STRIP(1, D, 0); } while (0)
    if (g.s == 2 && fused) TSII_DW_STRIP(2, 1, 1);
    else if (g.s == 2) TSII_DW_STRIP(2, 1, 0);
    else if (g.d == 2) TSII_DW_STRIP_D(2);
    else if (g.d == 4) TSII_DW_STRIP_D(4);
    else if (g.d == 8) TSII_DW_STRIP_D(8);
    else if (bb.y != nullptr) hipLaunchKernelGGL((dw_strip_kernel<1, 1, 2>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                                 sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out);
    else if (fused) hipLaunchKernelGGL((dw_strip_kernel<1, 1, 1>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                                       sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out);
    else hipLaunchKernelGGL((dw_strip_kernel<1, 1, 0>), grid, dim3(256), 0, st, in, pre, wT, bias, denom, keep, post_mul, g,
                            sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, out);
    return check_launch("dw_strip");
}

// ---- bf16 activation storage on the lean strip kernel (dw_lean_api.h; round 6) -----------------------------------------------
// The marching-column kernels of bf16_dw.hip request one row ahead at two waves per SIMD (231 VGPRs in the fused forms) and apply the
// producer's BatchNorm once per THREAD that loads an element (2-3x per element): 3.3 TB/s fused.  The lean kernel stages a slab
// once, applies the transform once per element and keeps a slab in flight across a whole step at three waves per SIMD -- and
// MEASURED SLOWER (round 6, tools/bf16_bench.py, profiles/r06f_bf16_bench_dw_*.log; 8 x 128^2 x 512, strip vs column kernel):
// forward + K6b 87.5 vs 81.4 us, plain forward 54.9 vs 49.8, dX + K6c 106.0 vs 110.4; 8 x 512^2 x 64 plain 115.6 vs 89.3; cfg 5's
// step 63.1 vs 60.4 ms.  The strip kernel is bound by its ELEMENT rate (LDS traffic and issue slots per element are those of the
// fp32 form, ~1.5e12 elements/s), which 4 bytes per element hide behind HBM and 2 bytes do not; its 64-byte pixel segments (4
// channels x 2 bytes per thread) also coalesce worse than the column kernels' 16-byte octets.  OFF in the stock library; the CPU
// suite keeps the form tested through an emulator-only switch.
#ifdef TSII_HIP_EMU
static int g_hdw_lean = 0;       // TEST-ONLY: 1 = the strip form, 0 = the marching-column kernels (the CPU suite runs both)
extern "C" void tsii_emu_set_hdw_lean(int v) { g_hdw_lean = v; }
#define HDW_LEAN g_hdw_lean
#else
#ifndef HDW_LEAN
#define HDW_LEAN 0               // A/B (tools/variants): 1 puts bf16 storage's stride-1 depth-wise layers on the strip kernel
#endif
#endif
bool hdw_lean_ok(int hout, int wout, int c, int s, int d) {
    if (!HDW_LEAN || !LS_ENABLE || s != 1 || c % 8 != 0 || hout < 1 || wout < 1) return false;
    if (d == 1) {
        const int64_t wmax = (int64_t)wout + 2, hmax = (int64_t)hout + 2;
        return c < (1 << 24) && hmax * wmax * 4 < (1ll << 24) && hmax * wmax * c * 4 < (1ll << 32) && (LS_ROWS * wmax + 64) * c * 4 < (1ll << 31);
    }
    if (!(d == 2 || d == 4) || wout < 8 * d || hout < 4 * d) return false;
    const int64_t wmax = (int64_t)wout + 2 * d, hmax = (int64_t)hout + 2 * d;
    return c < (1 << 24) && hmax * wmax * 4 < (1ll << 24) && hmax * wmax * c * 4 < (1ll << 32) && (LS_ROWS * wmax + 64) * c * 4 * d < (1ll << 31);
}
static FusedPlan hdw_lean_plan(int n, int hout, int wout, int c, int d) {
    FusedPlan fp;
    fp.rows_only = false;
    fp.phases = (unsigned)(d * d);
    fp.sp = plan_strip(n * d * d, cdiv(hout, d), cdiv(wout, d), c, 1, 1);
    return fp;
}
int64_t hdw_lean_rows(int n, int hout, int wout, int c, int s, int d) {
    if (!hdw_lean_ok(hout, wout, c, s, d) || (int64_t)n * d * d >= (1ll << 24)) return 0;
    const FusedPlan fp = hdw_lean_plan(n, hout, wout, c, d);
    return fp.sp.ok ? (int64_t)n * fused_rows_per_image(fp) : 0;
}
int launch_hdw_lean(const void* in, const float* w, const float* bias, int n, int hin, int win, int c, int d, int pad_h, int pad_w,
                    int hout, int wout, int flip, const float* in_sc, const float* in_sh, float in_neg, float in_hi, float* stats,
                    const void* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta,
                    float bn_eps, float bn_neg, float bn_hi, float* bn_part, void* out, hipStream_t st) {
    if (hdw_lean_rows(n, hout, wout, c, 1, d) <= 0) return 1;
    if (win > wout + 2 * d || hin > hout + 2 * d || pad_h < 0 || pad_w < 0) return 1;
    DtGeom g = {n, hin, win, c, 1, d, pad_h, pad_w, hout, wout, flip};
    const FusedPlan fp = hdw_lean_plan(n, hout, wout, c, d);
    const StripPlan sp = fp.sp;
    const int64_t nblk = (int64_t)sp.strips_x * sp.chunks_y * sp.cblocks * n * fp.phases;
    if (nblk >= (1ll << 31)) return 1;
    const dim3 grid((unsigned)nblk);
    const DwBN ib = {in_sc, in_sh, in_neg, in_hi};
    const DwBnBwd bb = {reinterpret_cast<const float*>(bn_y), bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, bn_neg, bn_hi, bn_part};
    const float* inf = reinterpret_cast<const float*>(in);
    float* outf = reinterpret_cast<float*>(out);
    const float* none = nullptr;
#define TSII_HDW_LEAN(MODE, DXE, PHV) hipLaunchKernelGGL((dw_lean_kernel<MODE, DXE, false, PHV, true>), grid, dim3(256), 0, st, inf, none, w, bias, none, none, none, g, \
                                                         sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, ib, stats, bb, outf)
    if (bn_y != nullptr) { if (d > 1) TSII_HDW_LEAN(2, true, true); else TSII_HDW_LEAN(2, true, false); }
    else if (in_sc != nullptr || stats != nullptr) { if (d > 1) TSII_HDW_LEAN(1, false, true); else TSII_HDW_LEAN(1, false, false); }
    else { if (d > 1) TSII_HDW_LEAN(0, false, true); else TSII_HDW_LEAN(0, false, false); }
#undef TSII_HDW_LEAN
    return check_launch("dw_lean (bf16 storage)");
}

// ---- marching-strip dW (stride 1): same ring of x rows as dw_strip_kernel, the 9 taps x 4 channels (+ bias)
// accumulate in registers over the whole strip; the next step's x rows AND dy pixels are in flight during the
// multiply-accumulate of the current one.  One partial row per block, combined by dw_reduce_kernel.
template <int S, int D>
__global__ __launch_bounds__(256, 2) void dw_strip_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                             const float* __restrict__ keep, const float* __restrict__ x,
                                                             const float* __restrict__ rmask, DtGeom g, int chunk_rows,
                                                             unsigned strips_x, unsigned chunks_y, unsigned cblocks, DwBN ib,
                                                             float* __restrict__ part) {
    constexpr int R = ST_R / S, TW = ST_TW / S;
    constexpr int PW = (TW - 1) * S + 2 * D + 1, NR = (R - 1) * S + 2 * D + 1;
    constexpr int NEW = R * S, PRO = NR - NEW;
    constexpr int CGS = ST_CB / 4, LANES = 256 / CGS, NP = R * TW / LANES;
    constexpr int PF = (NEW * PW + LANES - 1) / LANES;
    constexpr int NPX = R * TW;
    constexpr int TYS = LANES / TW;
    static_assert(PRO >= 0, "ring rows");
    constexpr int RING = NR * PW * ST_CB > 5 * 256 * 4 ? NR * PW * ST_CB : 5 * 256 * 4;   // also the final lane-combine buffer
    __shared__ __attribute__((aligned(16))) float ring[RING];
    __shared__ float planes[2][2][NPX];                        // keep / inv of a step's pixels, double buffered
    const unsigned cb = blockIdx.x % cblocks;
    unsigned b = blockIdx.x / cblocks;                         // partial row index
    const unsigned prow_idx = b;
    const unsigned sx = b % strips_x; b /= strips_x;
    const unsigned cy = b % chunks_y;
    const int64_t n = b / chunks_y;
    const int cg = threadIdx.x % CGS, lane = threadIdx.x / CGS;
    const int c = (int)cb * ST_CB + cg * 4;
    const bool cok = c < g.c;
    const int oy_beg = (int)cy * chunk_rows;
    const int oy_end = oy_beg + chunk_rows < g.hout ? oy_beg + chunk_rows : g.hout;
    const int ox0 = (int)sx * TW;
    const int iy_base = oy_beg * S - g.pad_h, ix0 = ox0 * S - g.pad_w;
    const int nsteps = (oy_end - oy_beg + R - 1) / R;

    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool bn_in = ib.sc != nullptr;
    if (bn_in && cok) { isc = *reinterpret_cast<const float4*>(ib.sc + c); ish = *reinterpret_cast<const float4*>(ib.sh + c); }

    float4 pf[PF];
    float pm[PF];
    auto fetch = [&](int rr0, int cnt) {
        const int iyb = iy_base + rr0;
        const int64_t pixbase = (n * g.hin + iyb) * (int64_t)g.win + ix0;
        const float* __restrict__ src = x + pixbase * g.c + c;
        const float* __restrict__ psrc = rmask != nullptr ? rmask + pixbase : nullptr;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            const int iy = iyb + row, ix = ix0 + px;
            pf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            pm[i] = 0.f;
            if (row < cnt && cok && iy >= 0 && iy < g.hin && ix >= 0 && ix < g.win) {
                const int off = row * g.win + px;
                pf[i] = *reinterpret_cast<const float4*>(src + off * g.c);
                pm[i] = psrc != nullptr ? psrc[off] : 1.f;
            }
        }
    };
    auto commit = [&](int rr0, int cnt) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int p = lane + LANES * i;
            const int row = p / PW, px = p - row * PW;
            if (row >= cnt) continue;
            float4 v = pf[i];
            const float m = pm[i];
            if (bn_in) {
                v.x = bn_act_load(v.x, isc.x, ish.x, ib.neg, ib.hi); v.y = bn_act_load(v.y, isc.y, ish.y, ib.neg, ib.hi);
                v.z = bn_act_load(v.z, isc.z, ish.z, ib.neg, ib.hi); v.w = bn_act_load(v.w, isc.w, ish.w, ib.neg, ib.hi);
            }
            v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            *reinterpret_cast<float4*>(ring + (((rr0 + row) % NR) * PW + px) * ST_CB + cg * 4) = v;
        }
    };
    const int tx = lane % TW, ty0 = lane / TW;
    const bool xok = cok && ox0 + tx < g.wout;
    // dy of this thread's 4 pixels for step s, and the keep / inv planes (threads 0..127: one pixel each)
    float4 gn[NP];
    float pl0 = 0.f, pl1 = 0.f;
    auto fetch_dy = [&](int s) {
        const int oyb = oy_beg + R * s;
        const float* __restrict__ src = dy + ((n * g.hout + oyb) * (int64_t)g.wout + ox0) * g.c + c;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int ty = ty0 + TYS * k;
            gn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xok && oyb + ty < oy_end) gn[k] = *reinterpret_cast<const float4*>(src + (ty * g.wout + tx) * g.c);
        }
        if (threadIdx.x < NPX) {
            const int oy = oyb + (int)threadIdx.x / TW, ox = ox0 + (int)threadIdx.x % TW;
            pl0 = 0.f; pl1 = 0.f;                              // out of range: no gradient
            if (oy < oy_end && ox < g.wout) {
                const int64_t q = (n * g.hout + oy) * (int64_t)g.wout + ox;
                pl0 = keep != nullptr ? keep[q] : 1.f;
                pl1 = inv != nullptr ? inv[q] : 1.f;
            }
        }
    };
    auto commit_planes = [&](int s) {
        if (threadIdx.x < NPX) { planes[s & 1][0][threadIdx.x] = pl0; planes[s & 1][1][threadIdx.x] = pl1; }
    };

    float4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    fetch_dy(0);
#pragma unroll
    for (int r = 0; r < PRO; r += NEW) {
        fetch(r, PRO - r < NEW ? PRO - r : NEW);
        commit(r, PRO - r < NEW ? PRO - r : NEW);
    }
    fetch(PRO, NEW);
    commit(PRO, NEW);
    commit_planes(0);
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        float4 gv[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) gv[k] = gn[k];
        if (more) { fetch(PRO + NEW * (s + 1), NEW); fetch_dy(s + 1); }
        const float* __restrict__ pls = &planes[s & 1][0][0];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            __builtin_amdgcn_sched_barrier(0);             // one pixel's 9 LDS reads in flight at a time (VGPR budget)
            const int ty = ty0 + TYS * k;
            const int pp = ty * TW + tx;
            if (pls[pp] == 0.f) continue;                  // hole / out of range: no gradient (partial_convolution.py:72)
            float4 gq = gv[k];
            acc[9].x += gq.x; acc[9].y += gq.y; acc[9].z += gq.z; acc[9].w += gq.w;   // bias: added after the division
            const float sc = pls[NPX + pp];
            gq.x *= sc; gq.y *= sc; gq.z *= sc; gq.w *= sc;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* rp = ring + ((((R * s + ty) * S + ky * D) % NR) * PW + tx * S) * ST_CB + cg * 4;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = *reinterpret_cast<const float4*>(rp + kx * D * ST_CB);
                    float4& a = acc[ky * 3 + kx];
                    a.x = fmaf(gq.x, v.x, a.x); a.y = fmaf(gq.y, v.y, a.y); a.z = fmaf(gq.z, v.z, a.z); a.w = fmaf(gq.w, v.w, a.w);
                }
            }
        }
        lds_barrier();
        if (more) { commit(PRO + NEW * (s + 1), NEW); commit_planes(s + 1); }
        lds_barrier();
    }
    // combine the 32 pixel lanes through the (free) ring: [10 taps][256 threads] float4, then 80 threads per ... sum
    float4* red4 = reinterpret_cast<float4*>(ring);
    float* prow = part + (int64_t)prow_idx * 10 * g.c;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 5; ++t) red4[t * 256 + threadIdx.x] = acc[half * 5 + t];
        __syncthreads();
        // 5 taps x 32 channels = 160 sums of 32 lanes each
        if (threadIdx.x < 160) {
            const int t = threadIdx.x / ST_CB, ch = threadIdx.x % ST_CB;
            if ((int)cb * ST_CB + ch < g.c) {
                const float* col = ring + (t * 256) * 4 + (ch / 4) * 4 + (ch % 4);
                float sum = 0.f;
#pragma unroll 8
                for (int l = 0; l < LANES; ++l) sum += col[l * CGS * 4];
                prow[(int64_t)(half * 5 + t) * g.c + (int)cb * ST_CB + ch] = sum;
            }
        }
    }
}

// sum the R partial rows (block = 32 columns x 8 row lanes, 4 loads in flight) and scatter back to the
// reference layout dw[c][t], db[c]
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ part, int R, int T, int C,
                                                        float* __restrict__ dwgt, float* __restrict__ dbias, int TP = 0) {
    __shared__ double sh[8][33];
    const int64_t len = (int64_t)(TP > 0 ? TP : T + 1) * C;    // TP: sub-rows of a partial row (default: the taps + the bias row)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t j = (int64_t)blockIdx.x * 32 + tx;
    double a[4] = {0, 0, 0, 0};
    if (j < len) {
        for (int r = ty; r < R; r += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + 8 * u;
                if (rr < R) a[u] += (double)part[(int64_t)rr * len + j];
            }
        }
    }
    sh[ty][tx] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (ty == 0 && j < len) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += sh[k][tx];
        const int t = (int)(j / C), c = (int)(j % C);
        if (t < T) dwgt[(int64_t)c * T + t] = (float)s;
        else if (dbias != nullptr) dbias[c] = (float)s;
    }
}

struct DwPlan {
    int W, CG, CGB, L, gx, gy, rpb, R;
};
static DwPlan plan_dw(int n, int ho, int c, bool vec) {
    DwPlan p;
    p.W = vec ? 4 : 1;
    p.CG = c / p.W;
    p.CGB = p.CG < 32 ? p.CG : 32;
    p.L = 256 / p.CGB;
    p.gx = cdiv(p.CG, p.CGB);
    const int rows_total = n * ho;
    int gy = 2048 / p.gx;
    if (gy < 1) gy = 1;
    if (gy > rows_total) gy = rows_total;
    p.rpb = cdiv(rows_total, gy);
    p.gy = cdiv(rows_total, p.rpb);
    p.R = p.gy;
    return p;
}

static int check_geom(const DwGeom& g, const char* who) {
    TSII_REQUIRE(g.n > 0 && g.h > 0 && g.w > 0 && g.c > 0 && g.kh > 0 && g.kw > 0 && g.sh > 0 && g.sw > 0 &&
                 g.dh > 0 && g.dw > 0 && g.ph >= 0 && g.pw >= 0, "%s: bad geometry", who);
    TSII_REQUIRE(g.ho == (g.h + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1 &&
                 g.wo == (g.w + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1,
                 "%s: output size %dx%d inconsistent with conv geometry", who, g.ho, g.wo);
    return 0;
}

}  // namespace tsii

using namespace tsii;

#define DW_GEOM() DwGeom g = {n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo}

static int dw_fwd_impl(const float* x, const float* rmask, const float* w, const float* bias,
                       const float* denom, const float* keep, int n, int h, int wd, int c, int kh, int kw,
                       int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo, DwBN ib, float* stats, float* y,
                       float* ws, void* stream) {
    TSII_REQUIRE(x && w && y && ws, "dw_fwd: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_fwd")) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_transpose(w, c, kh * kw, ws, st);  // [C][T] -> [T][C]
    if (rc) return rc;
    const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(ws);
    if (kh == 3 && kw == 3 && sh == sw && dh == dw) {   // marching-strip 3x3 stencil
        DtGeom tg = {n, h, wd, c, sh, dh, ph, pw, ho, wo, 0};
        rc = try_launch_dw_strip(x, rmask, ws, bias, denom, keep, nullptr, tg, y, st, ib, stats);
        if (rc <= 0) return rc;
    }
    TSII_REQUIRE(ib.sc == nullptr && stats == nullptr,
                 "dw_fwd_bn: the fused BatchNorm forms need the marching-strip path (tsii_dw_stat_rows() > 0, 16-byte aligned operands)");
    // measured on MI355X: the fully unrolled 3x3 form (more loads in flight) is SLOWER here -- these
    // stencils are bound by vector-memory instruction issue, not latency -- so it stays disabled
    const bool k3 = false;
    const int px = (vec && k3) ? DwPx<true>::value : DwPx<false>::value;
    const int64_t nblk = (int64_t)cdiv(cdiv(wo, px) * (vec ? c / 4 : c), 256) * ho * n;
    TSII_REQUIRE(nblk < (1ll << 31), "dw_fwd: grid limit");
    const dim3 grid((unsigned)nblk);
    if (vec && k3) hipLaunchKernelGGL((dw_fwd_kernel<4, true>), grid, dim3(256), 0, st, x, rmask, ws, bias, denom, keep, g, y);
    else if (vec) hipLaunchKernelGGL((dw_fwd_kernel<4, false>), grid, dim3(256), 0, st, x, rmask, ws, bias, denom, keep, g, y);
    else hipLaunchKernelGGL((dw_fwd_kernel<1, false>), grid, dim3(256), 0, st, x, rmask, ws, bias, denom, keep, g, y);
    return check_launch("dw_fwd");
}

extern "C" int tsii_dw_fwd(const float* x, const float* rmask, const float* w, const float* bias,
                           const float* denom, const float* keep, int n, int h, int wd, int c, int kh, int kw,
                           int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo, float* y, float* ws,
                           void* stream) {
    return dw_fwd_impl(x, rmask, w, bias, denom, keep, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, kNoDwBN, nullptr, y,
                       ws, stream);
}

extern "C" int64_t tsii_dw_stat_rows(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || c % 4 != 0) return 0;
    if (!(kh == 3 && kw == 3 && sh == sw && dh == dw) || !(dw_fused_ok(sh, dh) || dw_rows_dims_ok(ho, wo, c, sh, dh))) return 0;
    const FusedPlan fp = plan_fwd_strips(n, ho, wo, c, sh, dh);
    return (int64_t)n * fused_rows_per_image(fp);          // one partial row per strip chunk (of every phase)
}

// rows of the partials tsii_dw_bwd_dx_bn writes (0: that geometry has no fused form): strip chunks of the dX (= input) grid
extern "C" int64_t tsii_dw_bwd_stat_rows(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 4 != 0 || !(kh == 3 && kw == 3 && sh == sw && dh == dw)) return 0;
    FusedPlan fp;
    if (sh == 1 && (dh == 1 || dh == 2 || dh == 4 || dh == 8 || (ph == dh && pw == dw && dw_rows_dims_ok(h, wd, c, 1, dh)))) fp = plan_fwd_strips(n, h, wd, c, 1, dh);
    else if (sh == 2 && dh == 1 && ph == 1 && pw == 1) fp = plan_fwd_strips(n, h, wd, c, 1, 1);
    else return 0;
    return (int64_t)n * fused_rows_per_image(fp);
}

extern "C" int tsii_dw_fwd_bn(const float* x, const float* rmask, const float* w, const float* bias,
                              const float* denom, const float* keep, int n, int h, int wd, int c, int kh, int kw,
                              int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                              const float* in_scale, const float* in_shift, int in_act, float in_slope,
                              float* stat_part, float* y, float* ws, void* stream) {
    TSII_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dw_fwd_bn: in_scale / in_shift go together");
    DwBN ib;
    TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "dw_fwd_bn: activation %d has no load-time form", in_act);
    return dw_fwd_impl(x, rmask, w, bias, denom, keep, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, ib, stat_part, y, ws,
                       stream);
}

static int dw_bwd_dx_impl(const float* dy, const float* inv, const float* w, const float* rmask,
                          int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                          int dh, int dw, int ho, int wo, DwBnBwd bb, float* dx, float* ws, void* stream, float* dwpart = nullptr,
                          DwBN fold_ib = kNoDwBN) {
    TSII_REQUIRE(dy && w && dx && ws, "dw_bwd_dx: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_bwd_dx")) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_transpose(w, c, kh * kw, ws, st);
    if (rc) return rc;
    const bool vec = (c % 4 == 0) && aligned16(dy) && aligned16(dx) && aligned16(ws);
    if (kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == dw) {
        // stride 1: dx[i] = rmask[i] * sum_t w[t] * (dy*inv)[i + pad - t*d] -- the forward stencil with flipped taps
        DtGeom tg = {n, ho, wo, c, 1, dh, 2 * dh - ph, 2 * dw - pw, h, wd, 1};
        rc = try_launch_dw_strip(dy, inv, ws, nullptr, nullptr, nullptr, rmask, tg, dx, st, fold_ib, nullptr, bb, dwpart, fold_ib.sc != nullptr);
        if (rc <= 0) return rc;
    }
    TSII_REQUIRE(fold_ib.sc == nullptr || (sh == 2 && sw == 2), "dw_bwd_dxdw_bn2: this geometry has no form with the BatchNorm backward on load (tsii_dw_bwd_dxdw_fold_ok() == 0)");
    if (vec && kh == 3 && kw == 3 && sh == 2 && sw == 2 && dh == 1 && dw == 1 && ph == 1 && pw == 1) {   // marching strips
        const StripPlan sp = plan_strip(n, h, wd, c, 1, 1);     // strips of the input grid
        if (sp.ok) {
            const int64_t nblk2 = (int64_t)sp.strips_x * sp.chunks_y * sp.cblocks * n;
            if (bb.y != nullptr && dwpart != nullptr && fold_ib.sc != nullptr) {
                TSII_REQUIRE(aligned16(fold_ib.sc) && aligned16(fold_ib.sh), "dw_bwd_dxdw_bn2: the folded BatchNorm's operands must be 16-byte aligned");
                hipLaunchKernelGGL((dw_strip_dx2_kernel<true, true, true>), dim3((unsigned)nblk2), dim3(256), 0, st, dy, inv, ws, rmask, n, h, wd, c, ho, wo,
                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, bb, dx, dwpart, fold_ib);
            } else if (bb.y != nullptr && dwpart != nullptr)
                hipLaunchKernelGGL((dw_strip_dx2_kernel<true, true>), dim3((unsigned)nblk2), dim3(256), 0, st, dy, inv, ws, rmask, n, h, wd, c, ho, wo,
                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, bb, dx, dwpart);
            else if (bb.y != nullptr)
                hipLaunchKernelGGL((dw_strip_dx2_kernel<true, false>), dim3((unsigned)nblk2), dim3(256), 0, st, dy, inv, ws, rmask, n, h, wd, c, ho, wo,
                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, bb, dx, (float*)nullptr);
            else
                hipLaunchKernelGGL((dw_strip_dx2_kernel<false, false>), dim3((unsigned)nblk2), dim3(256), 0, st, dy, inv, ws, rmask, n, h, wd, c, ho, wo,
                                   sp.chunk_rows, sp.strips_x, sp.chunks_y, sp.cblocks, bb, dx, (float*)nullptr);
            return check_launch("dw_strip_dx2");
        }
    }
    TSII_REQUIRE(dwpart == nullptr, "dw_bwd_dxdw_bn: this geometry has no fused dX + dW form (tsii_dw_bwd_dxdw_ws_bytes() == 0)");
    TSII_REQUIRE(bb.y == nullptr, "dw_bwd_dx_bn: the BatchNorm-backward form needs a marching-strip path (tsii_dw_bwd_stat_rows() > 0)");
    const bool k3 = false;  // see dw_fwd
    const int px = (vec && k3) ? DwPx<true>::value : DwPx<false>::value;
    const int64_t nblk = (int64_t)cdiv(cdiv(wd, px) * (vec ? c / 4 : c), 256) * h * n;
    TSII_REQUIRE(nblk < (1ll << 31), "dw_bwd_dx: grid limit");
    const dim3 grid((unsigned)nblk);
    if (vec && k3) hipLaunchKernelGGL((dw_bwd_dx_kernel<4, true>), grid, dim3(256), 0, st, dy, inv, ws, rmask, g, dx);
    else if (vec) hipLaunchKernelGGL((dw_bwd_dx_kernel<4, false>), grid, dim3(256), 0, st, dy, inv, ws, rmask, g, dx);
    else hipLaunchKernelGGL((dw_bwd_dx_kernel<1, false>), grid, dim3(256), 0, st, dy, inv, ws, rmask, g, dx);
    return check_launch("dw_bwd_dx");
}

extern "C" int tsii_dw_bwd_dx(const float* dy, const float* inv, const float* w, const float* rmask,
                              int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* dx, float* ws, void* stream) {
    return dw_bwd_dx_impl(dy, inv, w, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, kNoBnBwd, dx, ws, stream);
}

extern "C" int tsii_dw_bwd_dx_bn(const float* dy, const float* inv, const float* w, const float* rmask,
                                 int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int ho, int wo,
                                 const float* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                 const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                 float* dx, float* bwd_part, float* ws, void* stream) {
    TSII_REQUIRE(bn_y && bn_mean && bn_var && bn_gamma && bn_beta && bwd_part, "dw_bwd_dx_bn: null pointer");
    TSII_REQUIRE(aligned16(bn_y) && aligned16(bn_mean) && aligned16(bn_var) && aligned16(bn_gamma) && aligned16(bn_beta),
                 "dw_bwd_dx_bn: BatchNorm operands must be 16-byte aligned");
    InBN tmp;
    TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "dw_bwd_dx_bn: activation %d has no load-time form", bn_act);
    const DwBnBwd bb = {bn_y, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, tmp.neg, tmp.hi, bwd_part};
    return dw_bwd_dx_impl(dy, inv, w, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, bb, dx, ws, stream);
}

// K6d: dX + K6c + the layer's weight gradient in ONE pass over (dy, y) -- the lean strip kernel's MODE 3 (dw_lean.h).
// -> bytes of the weight-gradient partial rows, 0 when the geometry has no such form (3x3, stride 1, dilation 1, c % 4 == 0, the
// sizes dw_lean_ok() takes); the K6c partial rows are those of tsii_dw_bwd_stat_rows().
extern "C" size_t tsii_dw_bwd_dxdw_ws_bytes(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 4 != 0 || !(kh == 3 && kw == 3 && dh == dw && dh >= 1)) return 0;
    if (sh == 2 && sw == 2 && ph == 1 && pw == 1 && dh == 1) {          // the stride-2 dX strips (dw_strip_dx2_kernel<true, true>)
        const StripPlan sp = plan_strip(n, h, wd, c, 1, 1);
        const int64_t rows = (int64_t)n * strip_rows_per_image(sp);
        if (!sp.ok || rows <= 0 || rows >= (1ll << 31) || (int64_t)sp.strips_x * sp.chunks_y * sp.cblocks * n >= (1ll << 31)) return 0;
        return (size_t)rows * 9 * (size_t)c * sizeof(float);
    }
    if (sh != 1 || sw != 1) return 0;
    const int d = dh;
    if (ph < 0 || pw < 0 || ph > 2 * d || pw > 2 * d) return 0;
    const int ho = h + 2 * ph - 2 * d, wo = wd + 2 * pw - 2 * d;
    if (ho <= 0 || wo <= 0) return 0;
    const DtGeom tg = {n, ho, wo, c, 1, d, 2 * d - ph, 2 * d - pw, h, wd, 1};
    if (!dw_dxdw_geom_ok(tg)) return 0;
    const FusedPlan fp = plan_fwd_strips(n, h, wd, c, 1, d);
    if (!fp.sp.ok || fp.phases != 1 || fp.rows_only) return 0;
    const int64_t rows = (int64_t)n * fused_rows_per_image(fp);
    if (rows <= 0 || rows >= (1ll << 31)) return 0;
    return (size_t)rows * 9 * (size_t)c * sizeof(float);
}

extern "C" int tsii_dw_bwd_dxdw_bn(const float* dy, const float* inv, const float* w, const float* rmask,
                                   int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                                   int dh, int dw, int ho, int wo,
                                   const float* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                   const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                   float* dx, float* bwd_part, float* dwgt, float* ws, void* ws_dw, size_t ws_dw_bytes, void* stream) {
    TSII_REQUIRE(bn_y && bn_mean && bn_var && bn_gamma && bn_beta && bwd_part && dwgt && ws_dw, "dw_bwd_dxdw_bn: null pointer");
    TSII_REQUIRE(aligned16(bn_y) && aligned16(bn_mean) && aligned16(bn_var) && aligned16(bn_gamma) && aligned16(bn_beta) && aligned16(ws_dw),
                 "dw_bwd_dxdw_bn: BatchNorm operands and the workspace must be 16-byte aligned");
    const size_t need = tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw);
    TSII_REQUIRE(need > 0, "dw_bwd_dxdw_bn: this geometry has no fused dX + dW form (tsii_dw_bwd_dxdw_ws_bytes() == 0)");
    TSII_REQUIRE(ws_dw_bytes >= need, "dw_bwd_dxdw_bn: weight-gradient workspace too small");
    InBN tmp;
    TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "dw_bwd_dxdw_bn: activation %d has no load-time form", bn_act);
    const DwBnBwd bb = {bn_y, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, tmp.neg, tmp.hi, bwd_part};
    int rc = dw_bwd_dx_impl(dy, inv, w, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, bb, dx, ws, stream, (float*)ws_dw);
    if (rc) return rc;
    const int rows = (int)(need / ((size_t)9 * c * sizeof(float)));
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)9 * c, 32)), dim3(256), 0, (hipStream_t)stream, (const float*)ws_dw, rows, 9, c, dwgt,
                       (float*)nullptr, 9);
    return check_launch("dw_reduce");
}

// K6e: tsii_dw_bwd_dxdw_bn whose incoming gradient is still the gradient w.r.t. the ACTIVATION of the BatchNorm that follows the
// layer (da2; bn2_y = the layer's raw output y2, bn2_coef = the [6][c] table tsii_bn_bwd_reduce leaves): that BatchNorm's backward
// is applied while the slab is staged (dw_lean.h MODE 4) -- 1 when the geometry has the form (3x3 / stride 1 / dilation 1).
extern "C" int tsii_dw_bwd_dxdw_fold_ok(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    // stride 1 / dilation 1 (the lean strip kernel) and stride 2 / padding 1 (the parity strips)
#ifndef DX2_FOLD
#define DX2_FOLD 1               // A/B (tools/variants): 0 = no K6e form on the stride-2 strips
#endif
    return (sh == sw && (sh == 1 || (DX2_FOLD && sh == 2)) && dh == 1 && dw == 1 && tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw) > 0) ? 1 : 0;
}

extern "C" int tsii_dw_bwd_dxdw_bn2(const float* da2, const float* bn2_y, const float* bn2_coef, int bn2_act, float bn2_slope,
                                    const float* inv, const float* w, const float* rmask,
                                    int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                                    int dh, int dw, int ho, int wo,
                                    const float* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                    const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                    float* dx, float* bwd_part, float* dwgt, float* ws, void* ws_dw, size_t ws_dw_bytes, void* stream) {
    TSII_REQUIRE(da2 && bn2_y && bn2_coef && bn_y && bn_mean && bn_var && bn_gamma && bn_beta && bwd_part && dwgt && ws_dw, "dw_bwd_dxdw_bn2: null pointer");
    TSII_REQUIRE(aligned16(bn_y) && aligned16(bn_mean) && aligned16(bn_var) && aligned16(bn_gamma) && aligned16(bn_beta) && aligned16(ws_dw) &&
                 aligned16(bn2_y) && aligned16(da2), "dw_bwd_dxdw_bn2: tensors and the workspace must be 16-byte aligned");
    TSII_REQUIRE(tsii_dw_bwd_dxdw_fold_ok(n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw), "dw_bwd_dxdw_bn2: this geometry has no form with the BatchNorm backward on load");
    const size_t need = tsii_dw_bwd_dxdw_ws_bytes(n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw);
    TSII_REQUIRE(ws_dw_bytes >= need, "dw_bwd_dxdw_bn2: weight-gradient workspace too small");
    InBN tmp, tmp2;
    TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "dw_bwd_dxdw_bn2: activation %d has no load-time form", bn_act);
    TSII_REQUIRE(make_in_bn(bn2_y, bn2_coef, bn2_act, bn2_slope, &tmp2) == 0, "dw_bwd_dxdw_bn2: activation %d has no load-time form", bn2_act);
    const DwBnBwd bb = {bn_y, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, tmp.neg, tmp.hi, bwd_part};
    const DwBN fold_ib = {bn2_y, bn2_coef, tmp2.neg, tmp2.hi};
    int rc = dw_bwd_dx_impl(da2, inv, w, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, bb, dx, ws, stream, (float*)ws_dw, fold_ib);
    if (rc) return rc;
    const int rows = (int)(need / ((size_t)9 * c * sizeof(float)));
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)9 * c, 32)), dim3(256), 0, (hipStream_t)stream, (const float*)ws_dw, rows, 9, c, dwgt,
                       (float*)nullptr, 9);
    return check_launch("dw_reduce");
}

extern "C" size_t tsii_dw_bwd_dw_ws_bytes(int n, int ho, int wo, int c, int kh, int kw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    // the scalar plan (taken when c % 4 != 0 or a pointer is unaligned) never needs more rows
    const DwPlan a = plan_dw(n, ho, c, c % 4 == 0), b = plan_dw(n, ho, c, false);
    int R = a.R > b.R ? a.R : b.R;
    if (kh == 3 && kw == 3) {
        for (int ss = 1; ss <= 2; ++ss) {       // strip plans of both strides (the stride is not part of this signature)
            const StripPlan sp = plan_strip(n, ho, wo, c, ss, 1);
            const int64_t g2 = (int64_t)n * sp.chunks_y * sp.strips_x;
            if (g2 > R) R = (int)g2;
        }
    }
    return (size_t)R * (size_t)(kh * kw + 1) * c * sizeof(float);
}

static int dw_bwd_dw_impl(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                          int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                          int dh, int dw, int ho, int wo, DwBN ib, float* dwgt, float* dbias, void* ws, size_t ws_bytes,
                          void* stream) {
    TSII_REQUIRE(dy && x && dwgt && ws, "dw_bwd_dw: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_bwd_dw")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, kh, kw), "dw_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && aligned16(dy) && aligned16(x) && (ib.sc == nullptr || (aligned16(ib.sc) && aligned16(ib.sh)));
    float* part = (float*)ws;
    if (vec && kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == dw && inv == nullptr && keep == nullptr && rmask == nullptr) {
        DtGeom rg = {n, h, wd, c, 1, dh, ph, pw, ho, wo, 0};
        if (dw_rows_ok(rg) && (int64_t)n * cdiv(c, DR_CB) < (1ll << 31)) {        // large dilation, mid-sized map, no planes: dw_rows.h
            const unsigned rcb = (unsigned)cdiv(c, DR_CB);
            if (ib.sc != nullptr) hipLaunchKernelGGL((dw_rows_dw_kernel<true>), dim3((unsigned)n * rcb), dim3(DR_THREADS), 0, st, dy, x, rg, rcb, ib, part);
            else hipLaunchKernelGGL((dw_rows_dw_kernel<false>), dim3((unsigned)n * rcb), dim3(DR_THREADS), 0, st, dy, x, rg, rcb, ib, part);
            int rc0 = check_launch("dw_rows_dw");
            if (rc0) return rc0;
            hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)10 * c, 32)), dim3(256), 0, st, part, n, 9, c, dwgt, dbias, 0);
            return check_launch("dw_reduce");
        }
    }
    if (vec && kh == 3 && kw == 3 && sh == sw && dh == dw &&
        ((sh == 1 && (dh == 1 || dh == 2 || dh == 4 || dh == 8)) || (sh == 2 && dh == 1))) {
        const StripPlan sp = plan_strip(n, ho, wo, c, sh, dh);   // marching strips
        if (sp.ok) {
            DtGeom tg = {n, h, wd, c, sh, dh, ph, pw, ho, wo, 0};
            const unsigned rows = (unsigned)n * sp.chunks_y * sp.strips_x;
            const dim3 grid(rows * sp.cblocks);
            if (sh == 2) hipLaunchKernelGGL((dw_strip_dw_kernel<2, 1>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, tg, sp.chunk_rows,
                                            sp.strips_x, sp.chunks_y, sp.cblocks, ib, part);
            else if (dh == 1) hipLaunchKernelGGL((dw_strip_dw_kernel<1, 1>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, tg, sp.chunk_rows,
                                                 sp.strips_x, sp.chunks_y, sp.cblocks, ib, part);
            else if (dh == 4) hipLaunchKernelGGL((dw_strip_dw_kernel<1, 4>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, tg, sp.chunk_rows,
                                                 sp.strips_x, sp.chunks_y, sp.cblocks, ib, part);
            else if (dh == 8) hipLaunchKernelGGL((dw_strip_dw_kernel<1, 8>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, tg, sp.chunk_rows,
                                                 sp.strips_x, sp.chunks_y, sp.cblocks, ib, part);
            else hipLaunchKernelGGL((dw_strip_dw_kernel<1, 2>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, tg, sp.chunk_rows,
                                    sp.strips_x, sp.chunks_y, sp.cblocks, ib, part);
            int rc0 = check_launch("dw_strip_dw");
            if (rc0) return rc0;
            hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)10 * c, 32)), dim3(256), 0, st, part, (int)rows, 9, c, dwgt, dbias, 0);
            return check_launch("dw_reduce");
        }
    }
    TSII_REQUIRE(ib.sc == nullptr, "dw_bwd_dw_bn: the fused BatchNorm form needs the marching-strip path (tsii_dw_stat_rows() > 0)");
    const DwPlan p = plan_dw(n, ho, c, vec);
    const dim3 grid((unsigned)(p.gx * p.gy));
    const bool k3 = (kh == 3 && kw == 3);
    if (vec && k3) hipLaunchKernelGGL((dw_bwd_dw_kernel<4, true>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, g, p.CGB, p.L, p.rpb, part);
    else if (vec) hipLaunchKernelGGL((dw_bwd_dw_kernel<4, false>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, g, p.CGB, p.L, p.rpb, part);
    else hipLaunchKernelGGL((dw_bwd_dw_kernel<1, false>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, g, p.CGB, p.L, p.rpb, part);
    int rc = check_launch("dw_bwd_dw");
    if (rc) return rc;
    const int T = kh * kw;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)(T + 1) * c, 32)), dim3(256), 0, st, part, p.R, T, c, dwgt, dbias, 0);
    return check_launch("dw_reduce");
}

extern "C" int tsii_dw_bwd_dw(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                              int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* dwgt, float* dbias, void* ws, size_t ws_bytes,
                              void* stream) {
    return dw_bwd_dw_impl(dy, inv, keep, x, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, kNoDwBN, dwgt, dbias, ws,
                          ws_bytes, stream);
}

extern "C" int tsii_dw_bwd_dw_bn(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                                 int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int ho, int wo,
                                 const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                 float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(in_scale && in_shift, "dw_bwd_dw_bn: null scale / shift");
    DwBN ib;
    TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "dw_bwd_dw_bn: activation %d has no load-time form", in_act);
    return dw_bwd_dw_impl(dy, inv, keep, x, rmask, n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, ib, dwgt, dbias, ws,
                          ws_bytes, stream);
}
