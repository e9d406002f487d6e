// K2 (round 4): 3x3 depth-wise stencils on SMALL maps (h * w <= 1024 pixels, stride 1, any dilation, "same" padding) with the whole
// map of a channel block in LDS.
//
// ImageFill's dilated levels (models/image_inpainting.py:29-33: dilation 2 / 4 / 8 on 32 x 32 maps, 1024 channels after expansion)
// ran on the marching-strip kernels: a 16-pixel strip with a dilation-8 halo is the full 32-pixel width, so the two strips of an
// image each read all of it (and 24 ring rows per 8 output rows on top): 0.20 ms per dilation-8 layer for 0.27 GB = 0.054 ms of
// traffic.  Here a block owns (image, 16 channels): it reads its 64 KB slice ONCE, 16 loads in
// flight per thread, and computes every output from LDS with per-tap bounds predicates -- no halo traffic at any dilation.  The
// two blocks that share the 128-byte lines of a pixel (channels 16 c .. 16 c + 31) carry consecutive logical ids, i.e. run on the
// same XCD at the same time (xcd_remap).
//   MODE 0 plain | 1 forward with BatchNorm on load and / or statistics partials (K6b) | 2 dX feeding a BatchNorm backward (K6c) |
//   3 (K6d, round 6) MODE 2 that also takes the layer's weight gradient: the 9 tap values of a pixel times the layer's input there
//   (act(bn(y)) * rmask) accumulate in 9 x 4 registers per thread over its 16 pixels; partial rows [9][C] in `stats`, strip layout;
//   DXE: dX epilogue (out = acc * post_mul, zero where post_mul == 0); PRE: staged input x per-pixel plane.
// Partial rows keep the strip plan's layout (the callers size them by tsii_dw_stat_rows / tsii_dw_bwd_stat_rows): an image's first
// row carries its sums, its other rows are written empty (count 0 / zeros).
// Included by dwconv.hip after dw_lean.h (f32x2 helpers, TSII_PIN_F2).
#pragma once

namespace tsii {

static constexpr int SM_CB = 16;                 // channels per block
static constexpr int SM_MAXPIX = 1024;           // pixels of a map
static constexpr int SM_THREADS = 512;           // 8 waves per block: two blocks (64 KB of LDS each) give a CU 16 waves
static constexpr int SM_LANES = SM_THREADS / 4;  // pixel lanes (4 channel quads per pixel)
static constexpr int SM_NP = SM_MAXPIX / SM_LANES;   // outputs per thread

#ifndef SM_MIN_DIL
#define SM_MIN_DIL 8             // smallest dilation sent here: measured on the chip (32 x 32 x 1024, fused forward) 0.131 ms per layer at any
#endif                           // dilation against 0.107 / 0.111 / 0.203 ms of the strips at dilation 2 / 4 / 8
#ifndef SM_ABL
#define SM_ABL 0                 // ablation builds: 1 = no output stores, 2 = no slab loads, 4 = no taps
#endif
#ifndef SM_ENABLE
#define SM_ENABLE 1              // A/B: 0 sends small maps back to the strip kernels
#endif
static inline bool dw_small_ok(const DtGeom& g) {
    return SM_ENABLE && g.s == 1 && g.d >= SM_MIN_DIL && g.hin == g.hout && g.win == g.wout && g.pad_h == g.d && g.pad_w == g.d &&
           (int64_t)g.hin * g.win <= SM_MAXPIX && g.c % 4 == 0 && (int64_t)g.hin * g.win * g.c * 4 < (1ll << 31);
}

template <int MODE, bool DXE, bool PRE>
__global__ __launch_bounds__(SM_THREADS, 2) void dw_small_kernel(
    const float* __restrict__ in, const float* __restrict__ pre, const float* __restrict__ wT, const float* __restrict__ bias,
    const float* __restrict__ denom, const float* __restrict__ keep, const float* __restrict__ post_mul, DtGeom g,
    unsigned cblocks, unsigned rows_per_image, DwBN ib, float* __restrict__ stats, DwBnBwd bb, float* __restrict__ out) {
    constexpr bool FUSED = (MODE == 1), BNB = (MODE >= 2), DWG = (MODE == 3);
    static_assert(!BNB || DXE, "K6c rides on the dX epilogue");
    static_assert(!FUSED || !DXE, "K6b rides on the forward epilogue");
    static_assert((SM_MAXPIX + 1) * SM_CB * sizeof(float) <= 160 * 1024, "the whole-map tile must fit gfx950's 160 KB of LDS per CU (65.6 KB: more than the 64 KB of earlier parts)");
    __shared__ __attribute__((aligned(16))) float tile[(SM_MAXPIX + 1) * SM_CB];      // [pixel][16 channels] + one pixel of zeros: what a tap outside the map reads
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned cb = b % cblocks;
    const int64_t n = b / cblocks;
    const int t = threadIdx.x;
    const int cq = t & 3, lane = t >> 2;
    const int C = g.c, H = g.hin, W = g.win, HW = H * W, D = g.d;
    const int c0 = (int)cb * SM_CB + cq * 4;
    const bool cok = c0 < C;
    const unsigned c0b = (unsigned)(cok ? c0 : C - 4) * 4u;
    const char* const ibase = reinterpret_cast<const char*>(in + n * HW * (int64_t)C);
    const float* const pbase = pre + (PRE ? n * HW : 0);
    auto opq = [](unsigned o) { TSII_OPAQUE_U32(o); return o; };

    // ---- the block's slice -> LDS (all loads first; BatchNorm + activation of the producer and the per-pixel plane at the store) ----
    f32x4 pf[SM_NP];
    float pm[SM_NP];
#pragma unroll
    for (int j = 0; j < SM_NP; ++j) {
        const int p = lane + SM_LANES * j;
        const unsigned pc = (unsigned)(p < HW ? p : HW - 1);
        pf[j] = (SM_ABL & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(ibase + opq(__umul24(pc * 4u, (unsigned)C) + c0b));
        pm[j] = PRE ? pbase[pc] : 1.f;
    }
    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = cok ? *reinterpret_cast<const f32x4*>(wT + (g.flip ? 8 - k : k) * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f}, bq = {0.f, 0.f, 0.f, 0.f};
    const bool has_bn = FUSED && ib.sc != nullptr;
    if (has_bn && cok) { isc = *reinterpret_cast<const f32x4*>(ib.sc + c0); ish = *reinterpret_cast<const f32x4*>(ib.sh + c0); }
    if (!DXE && bias != nullptr && cok) bq = *reinterpret_cast<const f32x4*>(bias + c0);
    const float bn_neg = has_bn ? ib.neg : 1.f;
    const bool hi_finite = has_bn && ib.hi < __builtin_huge_valf();
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bga = bmu, bbe = bmu;
    if (BNB && cok) {
        bmu = *reinterpret_cast<const f32x4*>(bb.mean + c0);
        const f32x4 vr = *reinterpret_cast<const f32x4*>(bb.var + c0);
        bis = f32x4{1.0f / sqrtf(vr.x + bb.eps), 1.0f / sqrtf(vr.y + bb.eps), 1.0f / sqrtf(vr.z + bb.eps), 1.0f / sqrtf(vr.w + bb.eps)};
        bga = *reinterpret_cast<const f32x4*>(bb.gamma + c0);
        bbe = *reinterpret_cast<const f32x4*>(bb.beta + c0);
    }
    char* const lthr = reinterpret_cast<char*>(tile) + t * 16;
#pragma unroll
    for (int j = 0; j < SM_NP; ++j) {
        f32x4 v = pf[j];
        if (FUSED) {
            f32x2 z0 = fma2(v.xy, isc.xy, ish.xy), z1 = fma2(v.zw, isc.zw, ish.zw);
            z0 = max2(z0, z0 * bn_neg); z1 = max2(z1, z1 * bn_neg);
            if (hi_finite) { z0 = min2(z0, f32x2{ib.hi, ib.hi}); z1 = min2(z1, f32x2{ib.hi, ib.hi}); }
            v = cat4(z0, z1);
        }
        v *= pm[j];
        if (lane + SM_LANES * j < HW) *reinterpret_cast<f32x4*>(lthr + SM_THREADS * 16 * j) = v;       // pixel p, quad cq at byte (p * 4 + cq) * 16
    }
    if (t < 4) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(tile) + (SM_MAXPIX * 4 + t) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // ---- outputs: pixel p = lane + SM_LANES j; taps outside the map are zeros (selected, never multiplied) ---------------------------
    const char* const rthr = reinterpret_cast<const char*>(tile) + cq * 16;
    char* const obase = reinterpret_cast<char*>(out + n * HW * (int64_t)C);
    const char* const ybase = reinterpret_cast<const char*>(BNB ? bb.y + n * HW * (int64_t)C : nullptr);
    const float* const dpl = denom != nullptr ? denom + n * HW : nullptr;
    const float* const kpl = keep != nullptr ? keep + n * HW : nullptr;
    const float* const qpl = post_mul != nullptr ? post_mul + n * HW : nullptr;
    f32x4 P = {0.f, 0.f, 0.f, 0.f};
    f32x2 va[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    int cnt = 0;
    f32x4 dwa[DWG ? 9 : 1];                                        // K6d: weight-gradient accumulators per (flipped) window tap
#pragma unroll
    for (int k = 0; k < (DWG ? 9 : 1); ++k) dwa[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned wmagic = (unsigned)(((1 << 20) + W - 1) / W);    // p / W for p < 1024, W <= 1024: (p * magic) >> 20 is exact
#pragma unroll 2
    for (int j = 0; j < SM_NP; ++j) {
        const int p = lane + SM_LANES * j;
        if (j * SM_LANES >= HW) break;                            // block-uniform
        const bool pok = p < HW && cok;
        const int pc = p < HW ? p : HW - 1;
        const int y = (int)(((unsigned)pc * wmagic) >> 20), x = pc - y * W;
        float e0 = 1.f, e1 = 1.f;                                  // epilogue planes, requested before the taps
        if (DXE) e0 = qpl != nullptr ? qpl[pc] : 1.f;
        else { e0 = dpl != nullptr ? dpl[pc] : 1.f; e1 = kpl != nullptr ? kpl[pc] : 1.f; }
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (BNB) yv = *reinterpret_cast<const f32x4*>(ybase + opq(__umul24((unsigned)pc * 4u, (unsigned)C) + c0b));
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        f32x4 ein = {0.f, 0.f, 0.f, 0.f};                          // K6d: the layer's input at this pixel, act(bn(y)) * rmask
        if (DWG && pok && e0 != 0.f) {
            const f32x2 g0 = fma2((yv.xy - bmu.xy) * bis.xy, bga.xy, bbe.xy), g1 = fma2((yv.zw - bmu.zw) * bis.zw, bga.zw, bbe.zw);
            const f32x2 q0 = min2(max2(g0, g0 * bb.neg), f32x2{bb.hi, bb.hi}) * e0, q1 = min2(max2(g1, g1 * bb.neg), f32x2{bb.hi, bb.hi}) * e0;
            ein = cat4(q0, q1);
        }
        // a tap outside the map reads the zero pixel (one select on the index, none on the data)
        int qrow[3], qcol[3];
        bool rok[3], cokx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int yy = y + (k - 1) * D, xx = x + (k - 1) * D;
            rok[k] = (unsigned)yy < (unsigned)H; cokx[k] = (unsigned)xx < (unsigned)W;
            qrow[k] = yy * W; qcol[k] = xx;
        }
#pragma unroll
        for (int ky = 0; ky < ((SM_ABL & 4) ? 1 : 3); ++ky) {
#pragma unroll
            for (int kx = 0; kx < ((SM_ABL & 4) ? 1 : 3); ++kx) {
                const int q = (rok[ky] && cokx[kx]) ? qrow[ky] + qcol[kx] : SM_MAXPIX;
                const f32x4 v = *reinterpret_cast<const f32x4*>(rthr + q * (SM_CB * 4));
                a0 = fma2(v.xy, w[ky * 3 + kx].xy, a0);
                a1 = fma2(v.zw, w[ky * 3 + kx].zw, a1);
                if (DWG) {
                    f32x4& q = dwa[DWG ? ky * 3 + kx : 0];
                    q = cat4(fma2(v.xy, ein.xy, q.xy), fma2(v.zw, ein.zw, q.zw));
                }
            }
        }
        if (DXE) {
            a0 *= e0; a1 *= e0;
            if (e0 == 0.f) { a0 = f32x2{0.f, 0.f}; a1 = a0; }
        } else {
            const float r = 1.0f / e0;
            a0 *= r; a1 *= r;
            a0 += bq.xy; a1 += bq.zw;
            if (e1 == 0.f) { a0 = f32x2{0.f, 0.f}; a1 = a0; }
        }
        if (pok) {
            if (!(SM_ABL & 1) || a0.x == 12345.f) __builtin_nontemporal_store(cat4(a0, a1), reinterpret_cast<f32x4*>(obase + opq(__umul24((unsigned)p * 4u, (unsigned)C) + (unsigned)c0 * 4u)));
            if (FUSED) {
                if (cnt == 0) P = cat4(a0, a1);
                const f32x2 d0 = a0 - P.xy, d1 = a1 - P.zw;
                va[0] += d0; va[1] += d1;
                va[2] = fma2(d0, d0, va[2]); va[3] = fma2(d1, d1, va[3]);
                ++cnt;
            }
            if (BNB) {
                const f32x2 h0 = (yv.xy - bmu.xy) * bis.xy, h1 = (yv.zw - bmu.zw) * bis.zw;
                const f32x2 z0 = fma2(h0, bga.xy, bbe.xy), z1 = fma2(h1, bga.zw, bbe.zw);
                f32x2 d0 = a0, d1 = a1;
                d0.x *= (z0.x > 0.f && z0.x < bb.hi) ? 1.f : (z0.x > 0.f ? 0.f : bb.neg);
                d0.y *= (z0.y > 0.f && z0.y < bb.hi) ? 1.f : (z0.y > 0.f ? 0.f : bb.neg);
                d1.x *= (z1.x > 0.f && z1.x < bb.hi) ? 1.f : (z1.x > 0.f ? 0.f : bb.neg);
                d1.y *= (z1.y > 0.f && z1.y < bb.hi) ? 1.f : (z1.y > 0.f ? 0.f : bb.neg);
                va[0] += d0; va[1] += d1;
                va[2] = fma2(d0, h0, va[2]); va[3] = fma2(d1, h1, va[3]);
            }
        }
    }
    if (!(BNB || (FUSED && stats != nullptr))) return;
    // ---- merge the pixel lanes of every channel through the (now free) tile ----------------------------------------------------
    __syncthreads();
    float* mrg = tile;                                   // [512][13]
    float* mt = mrg + t * 13;
    mt[0] = (float)cnt;
    mt[1] = P.x; mt[2] = P.y; mt[3] = P.z; mt[4] = P.w;
    mt[5] = va[0].x; mt[6] = va[0].y; mt[7] = va[1].x; mt[8] = va[1].y;
    mt[9] = va[2].x; mt[10] = va[2].y; mt[11] = va[3].x; mt[12] = va[3].y;
    __syncthreads();
    if (BNB) {
        if (t < 2 * SM_CB) {
            const int which = t / SM_CB, ch = t % SM_CB;
            if ((int)cb * SM_CB + ch < C) {
                float sum = 0.f;
                for (int l = 0; l < SM_LANES; ++l) sum += mrg[(l * 4 + ch / 4) * 13 + 5 + which * 4 + ch % 4];
                bb.part[(n * rows_per_image * 2 + which) * C + (int)cb * SM_CB + ch] = sum;
                for (unsigned r = 1; r < rows_per_image; ++r) bb.part[((n * rows_per_image + r) * 2 + which) * C + (int)cb * SM_CB + ch] = 0.f;
            }
        }
        if (DWG) {
            // K6d: the 128 pixel lanes of every channel, 5 + 4 taps at a time through the tile ([5][512] float4 = 40 KB)
            static_assert(5 * SM_THREADS * 16 <= (SM_MAXPIX + 1) * SM_CB * 4, "weight-gradient merge buffer fits the tile");
            f32x4* m4 = reinterpret_cast<f32x4*>(tile);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (half * 5 + k < 9) m4[k * SM_THREADS + t] = dwa[half * 5 + k < 9 ? half * 5 + k : 0];
                __syncthreads();
                const int nt = half == 0 ? 5 : 4;
                if (t < nt * SM_CB) {
                    const int k = t / SM_CB, ch = t % SM_CB;
                    if ((int)cb * SM_CB + ch < C) {
                        const float* col = tile + (k * SM_THREADS + ch / 4) * 4 + ch % 4;
                        float sum = 0.f;
#pragma unroll 8
                        for (int l = 0; l < SM_LANES; ++l) sum += col[l * 16];
                        const int kk = half * 5 + k, tap = g.flip ? 8 - kk : kk;
                        stats[((n * rows_per_image) * 9 + tap) * C + (int)cb * SM_CB + ch] = sum;
                        for (unsigned r = 1; r < rows_per_image; ++r) stats[((n * rows_per_image + r) * 9 + tap) * C + (int)cb * SM_CB + ch] = 0.f;
                    }
                }
            }
        }
    } else if (t < SM_CB && (int)cb * SM_CB + t < C) {
        // (count, pivot, s1, s2) per thread re-based to a common pivot:  s1' = s1 + n dp,  s2' = s2 + 2 dp s1 + n dp^2
        const int ch = t, mcg = ch / 4, mi = ch % 4;
        float nn = 0.f, pv = 0.f, s1 = 0.f, s2 = 0.f;
        bool have = false;
        {   // common pivot: an interior lane's (lane 33) when it saw pixels -- pixel 0 is a corner, the typical outlier of a channel
            const float* qi = mrg + (33 * 4 + mcg) * 13;
            if (qi[0] != 0.f) { pv = qi[1 + mi]; have = true; }
        }
        for (int l = 0; l < SM_LANES; ++l) {
            const float* q = mrg + (l * 4 + mcg) * 13;
            const float n_t = q[0];
            if (n_t == 0.f) continue;
            if (!have) { pv = q[1 + mi]; have = true; }
            const float dp = q[1 + mi] - pv, b1 = q[5 + mi], b2 = q[9 + mi];
            s1 += fmaf(n_t, dp, b1);
            s2 += b2 + dp * (2.f * b1 + n_t * dp);
            nn += n_t;
        }
        float* sp = stats + n * rows_per_image * 4 * C + (int)cb * SM_CB + ch;
        sp[0] = nn;
        sp[C] = pv;
        sp[2 * (int64_t)C] = s1;
        sp[3 * (int64_t)C] = s2;
        for (unsigned r = 1; r < rows_per_image; ++r) sp[(int64_t)r * 4 * C] = 0.f;      // empty rows: count 0
    }
}

}  // namespace tsii
