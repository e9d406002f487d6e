// K2 (round 4): the stride-2 forward marching strip in the lean form of dw_lean.h.
//
// The round-3 stride-2 strip (dw_strip_kernel<2, 1, *>, dwconv.hip) shares the generic ring / index machinery of the stride-1 kernel it
// was derived from: 231 VGPRs in its BatchNorm-fused form (2 waves per SIMD; a build capped at 3 waves spills 102 registers), its
// stride-2 column reads walk LDS at 256-byte steps (every pixel of a wave on the same 32 banks: 27-32 % of its LDS-active cycles are
// bank conflicts, profiles/r03_pmc_sq_bs32.csv), and it ran the fused forward at 3.0-3.9 TB/s.  Same recipe as dw_lean.h:
//   * two plain LDS buffers of 9 input rows (8 new rows per step + the one row the previous step shares), static offsets, one
//     LDS-only barrier per step, the carried row copied LDS -> LDS into the buffer nobody reads;
//   * even and odd input columns sit in separate planes of a buffer row, so the stride-2 taps of neighbouring outputs are
//     NEIGHBOURING 128-byte pixels again (conflict-free like the stride-1 strip);
//   * one output pixel per thread (4 rows x 8 columns x 8 channel quads per step): 9 ds_read_b128 feed 36 FMAs -- a stride-2 layer
//     moves four input pixels per output, so the arithmetic is a small fraction of the slab traffic;
//   * slab loads, plane handling, BatchNorm-on-load and the statistics partials (K6b) exactly as in the stride-1 kernel.
// Forward only (the stride-2 dX has its own kernel).  Included by dwconv.hip after dw_lean.h.
#pragma once

namespace tsii {

static constexpr int L2_OR = 4, L2_OTW = 8;                    // output rows x columns per step
static constexpr int L2_IN = 8;                                // new input rows per step
static constexpr int L2_PW = (L2_OTW - 1) * 2 + 3;             // 17 input columns per strip
static constexpr int L2_PWH = (L2_PW + 1) / 2;                 // 9 pixel slots per column parity
static constexpr int L2_ROWS = L2_IN + 1;                      // 9 buffer rows
static constexpr int L2_ROWB = 2 * L2_PWH * LS_PIXB;           // bytes of one buffer row (2304)
static constexpr int L2_BUFB = L2_ROWS * L2_ROWB;              // bytes per buffer (20736)
static constexpr int L2_PF = (L2_IN * L2_PW + 31) / 32;        // slab pixels per thread per step (5)
static constexpr int L2_NPX = L2_OR * L2_OTW;                  // 32 outputs per channel quad and step

// MODE 0: plain; 1: BatchNorm on load and / or statistics partials (K6b).  PRE: the staged input is multiplied by a per-pixel plane.
template <int MODE, bool PRE>
__global__ __launch_bounds__(256, 3) void dw_lean_s2_kernel(
    const float* __restrict__ in, const float* __restrict__ pre, const float* __restrict__ wT, const float* __restrict__ bias,
    const float* __restrict__ denom, const float* __restrict__ keep, DtGeom g, int chunk_rows,
    unsigned strips_x, unsigned chunks_y, unsigned cblocks, DwBN ib, float* __restrict__ stats, float* __restrict__ out) {
    constexpr bool FUSED = (MODE == 1);
    __shared__ __attribute__((aligned(16))) float lbuf[2 * L2_BUFB / 4];
    __shared__ __attribute__((aligned(8))) float lplanes[2][L2_NPX][2];
    __shared__ __attribute__((aligned(16))) float lconst[3][LS_CB];      // scale, shift, bias
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned sx = b % strips_x; b /= strips_x;
    const unsigned cb = b % cblocks; b /= cblocks;
    const unsigned cy = b % chunks_y;
    const int64_t n = b / chunks_y;
    const int t = threadIdx.x;
    const int cg = t & 7, lane = t >> 3;
    const int C = g.c;
    const int c0 = (int)cb * LS_CB + cg * 4;
    const bool cok = c0 < C;
    const int oy_beg = (int)cy * chunk_rows;
    const int oy_end = oy_beg + chunk_rows < g.hout ? oy_beg + chunk_rows : g.hout;
    const int ox0 = (int)sx * L2_OTW;
    const int iy_base = 2 * oy_beg - g.pad_h, ix0 = 2 * ox0 - g.pad_w;   // input row of buffer row 0 at step 0 / input column of slab column 0
    const int nsteps = (oy_end - oy_beg + L2_OR - 1) / L2_OR;
    const bool col_interior = ix0 >= 0 && ix0 + L2_PW <= g.win;
    const float bn_neg = FUSED && ib.sc != nullptr ? ib.neg : 1.f;
    const bool hi_finite = FUSED && ib.sc != nullptr && ib.hi < __builtin_huge_valf();

    if (t < LS_CB) {
        const int ch = (int)cb * LS_CB + t;
        const bool ok = ch < C;
        lconst[0][t] = (FUSED && ib.sc != nullptr && ok) ? ib.sc[ch] : 1.f;
        lconst[1][t] = (FUSED && ib.sc != nullptr && ok) ? ib.sh[ch] : 0.f;
        lconst[2][t] = (bias != nullptr && ok) ? bias[ch] : 0.f;
    }
    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cok) {
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(wT + k * C + c0);
    }

    // pixel slot of (buffer row, slab column): columns of one parity are neighbours
    auto slot = [](int row, int col) { return (row * 2 + (col & 1)) * L2_PWH + (col >> 1); };
    // ---- slab staging: 8 rows x 17 pixels per step, pixel p = lane + 32 i of thread t (all loads unconditional, see dw_lean.h) ----
    unsigned poff[L2_PF], loff[L2_PF];
    unsigned rowpk = 0, pxpk = 0;
#pragma unroll
    for (int i = 0; i < L2_PF; ++i) {
        const int pr = lane + 32 * i;
        const int p = pr < L2_IN * L2_PW ? pr : L2_IN * L2_PW - 1;
        const int row = p / L2_PW, px = p - row * L2_PW;
        poff[i] = (unsigned)(row * g.win + px) * 4u;
        loff[i] = (unsigned)(slot(row + 1, px) * LS_PIXB + cg * 16);
        rowpk |= (unsigned)row << (4 * i);
        pxpk |= (unsigned)px << (5 * i);
    }
    const bool item4 = lane < L2_IN * L2_PW - 128;             // item 4 exists for the first 8 pixel lanes only
    const unsigned c0b = (unsigned)(cok ? c0 : C - 4) * 4u;
    auto opq = [](unsigned o) { TSII_OPAQUE_U32(o); return o; };

    f32x4 pf[L2_PF];
    float pm[L2_PF];
    unsigned vmask = 0;
    const int64_t img_pix = n * g.hin * (int64_t)g.win;
    const char* const ibase = reinterpret_cast<const char*>(in + img_pix * C);
    const char* const ipre = reinterpret_cast<const char*>(pre + img_pix);
    int iyb = iy_base + 1;
    const char* sb = reinterpret_cast<const char*>(in + (img_pix + (int64_t)iyb * g.win + ix0) * C);
    const char* pb = reinterpret_cast<const char*>(pre + (img_pix + (int64_t)iyb * g.win + ix0));
    const int64_t sb_step = (int64_t)L2_IN * g.win * C * 4, pb_step = (int64_t)L2_IN * g.win * 4;
    auto fetch = [&]() {                                      // global -> registers, slab rows iyb .. iyb + 7
        const bool interior = col_interior && iyb >= 0 && iyb + L2_IN <= g.hin;
        unsigned po[L2_PF];
        vmask = 31u;
#pragma unroll
        for (int i = 0; i < L2_PF; ++i) po[i] = poff[i];
        if (!interior) {
            vmask = 0;
#pragma unroll
            for (int i = 0; i < L2_PF; ++i) {
                const int iy = iyb + (int)((rowpk >> (4 * i)) & 15u), ix = ix0 + (int)((pxpk >> (5 * i)) & 31u);
                if ((unsigned)iy < (unsigned)g.hin && (unsigned)ix < (unsigned)g.win) vmask |= 1u << i;
                const int iyc = iy < 0 ? 0 : (iy >= g.hin ? g.hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= g.win ? g.win - 1 : ix);
                po[i] = (unsigned)(iyc * g.win + ixc) * 4u;
            }
        }
        const char* const ab = interior ? sb : ibase;
        const char* const mb = interior ? pb : ipre;
#pragma unroll
        for (int i = 0; i < L2_PF; ++i) {
            pf[i] = *reinterpret_cast<const f32x4*>(ab + opq(__umul24(po[i], (unsigned)C) + c0b));
            pm[i] = PRE ? *reinterpret_cast<const float*>(mb + opq(po[i])) : 1.f;
        }
        iyb += L2_IN; sb += sb_step; pb += pb_step;
    };
    char* const lb0 = reinterpret_cast<char*>(lbuf);
    const char* const cthr = reinterpret_cast<const char*>(&lconst[0][0]) + cg * 16;
    auto stage = [&](f32x4 v, float m, bool inside, const f32x4& isc, const f32x4& ish) {
        if (FUSED) {
            f32x2 z0 = fma2(v.xy, isc.xy, ish.xy), z1 = fma2(v.zw, isc.zw, ish.zw);
            z0 = max2(z0, z0 * bn_neg); z1 = max2(z1, z1 * bn_neg);
            if (hi_finite) { z0 = min2(z0, f32x2{ib.hi, ib.hi}); z1 = min2(z1, f32x2{ib.hi, ib.hi}); }
            v = cat4(z0, z1);
        }
        v *= m;
        if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};
        return v;
    };
    auto commit_slab = [&](int which, const f32x4& isc, const f32x4& ish) {     // registers -> buffer `which` rows 1..8
        char* const T = lb0 + which * L2_BUFB;
#pragma unroll
        for (int i = 0; i < L2_PF; ++i) {
            const f32x4 v = stage(pf[i], pm[i], vmask == 31u ? true : (bool)((vmask >> i) & 1u), isc, ish);
            if (i < 4 || item4) *reinterpret_cast<f32x4*>(T + loff[i]) = v;
        }
    };
    auto commit = [&](int which) {                            // row 0 of buffer `which` = row 8 of the other buffer, then the slab
        char* const T = lb0 + which * L2_BUFB + t * 16;
        const char* const S = lb0 + (which ^ 1) * L2_BUFB + L2_IN * L2_ROWB + t * 16;
        f32x4 cv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < L2_ROWB / 16) cv = *reinterpret_cast<const f32x4*>(S);
        f32x4 isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f};
        if (FUSED) { isc = *reinterpret_cast<const f32x4*>(cthr); ish = *reinterpret_cast<const f32x4*>(cthr + LS_PIXB); }
        if (t < L2_ROWB / 16) *reinterpret_cast<f32x4*>(T) = cv;
        commit_slab(which, isc, ish);
    };
    float pl0 = 1.f, pl1 = 1.f;
    auto fetch_planes = [&](int s) {
        const int tp = t & (L2_NPX - 1);
        const int oy = oy_beg + L2_OR * s + tp / L2_OTW, ox = ox0 + tp % L2_OTW;
        const bool ok = oy < oy_end && ox < g.wout;
        const int64_t q = (n * g.hout + (ok ? oy : oy_beg)) * (int64_t)g.wout + (ok ? ox : ox0);
        pl0 = denom != nullptr ? denom[q] : 1.f;
        pl1 = keep != nullptr ? keep[q] : 1.f;
    };
    auto commit_planes = [&](int s) {
        const float r0 = 1.0f / pl0, r1 = pl1;               // one IEEE division per pixel
        if (t < L2_NPX) {
            lplanes[s & 1][t][0] = r0;
            lplanes[s & 1][t][1] = r1;
        }
    };

    // ---- prologue: the input row above the first step -> buffer 0 row 0, slab 0 -> rows 1..8 ------------------------------
    fetch_planes(0);
    __syncthreads();                                           // lconst
    {
        const f32x4 isc = *reinterpret_cast<const f32x4*>(cthr), ish = *reinterpret_cast<const f32x4*>(cthr + LS_PIXB);
        const int px = lane < L2_PW ? lane : L2_PW - 1;
        const int iy = iy_base, ix = ix0 + px;
        const bool inside = (unsigned)iy < (unsigned)g.hin && (unsigned)ix < (unsigned)g.win;
        const int iyc = iy < 0 ? 0 : (iy >= g.hin ? g.hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= g.win ? g.win - 1 : ix);
        const unsigned q = (unsigned)(iyc * g.win + ixc) * 4u;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ibase + opq(__umul24(q, (unsigned)C) + c0b));
        const float m = PRE ? *reinterpret_cast<const float*>(ipre + opq(q)) : 1.f;
        const f32x4 sv = stage(v, m, inside, isc, ish);
        if (lane < L2_PW) *reinterpret_cast<f32x4*>(lb0 + slot(0, px) * LS_PIXB + cg * 16) = sv;
        fetch();
        commit_slab(0, isc, ish);
    }
    commit_planes(0);
    if (nsteps > 1) { fetch(); fetch_planes(1); }
    __syncthreads();

    // this thread's output pixel of a step: row ty, column tx
    const int tx = lane & 7, ty = lane >> 3;
    const bool xok = cok && ox0 + tx < g.wout;
    const unsigned orow = (unsigned)g.wout * (unsigned)C * 4u;
    const unsigned ooff = (unsigned)ty * orow + ((unsigned)tx * (unsigned)C + (unsigned)c0) * 4u;
    const char* const rthr = lb0 + slot(2 * ty, 2 * tx) * LS_PIXB + cg * 16;
    char* ob = reinterpret_cast<char*>(out + ((n * g.hout + oy_beg) * (int64_t)g.wout + ox0) * C);
    const int64_t ob_step = (int64_t)L2_OR * g.wout * C * 4;

    f32x4 P = {0.f, 0.f, 0.f, 0.f};
    f32x2 va[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    auto step = [&](int s, bool more, bool more2) {
        const int rows_left = oy_end - (oy_beg + L2_OR * s);
        const char* const rb = rthr + (s & 1) * L2_BUFB;
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // columns 2 tx, 2 tx + 1, 2 tx + 2: even plane slot tx, odd plane slot tx, even plane slot tx + 1
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(rb + ky * L2_ROWB);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(rb + ky * L2_ROWB + L2_PWH * LS_PIXB);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(rb + ky * L2_ROWB + LS_PIXB);
            a0 = fma2(v0.xy, w[ky * 3 + 0].xy, a0); a1 = fma2(v0.zw, w[ky * 3 + 0].zw, a1);
            a0 = fma2(v1.xy, w[ky * 3 + 1].xy, a0); a1 = fma2(v1.zw, w[ky * 3 + 1].zw, a1);
            a0 = fma2(v2.xy, w[ky * 3 + 2].xy, a0); a1 = fma2(v2.zw, w[ky * 3 + 2].zw, a1);
        }
        const f32x2 pq = *reinterpret_cast<const f32x2*>(&lplanes[s & 1][ty * L2_OTW + tx][0]);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(cthr + 2 * LS_PIXB);
        a0 *= pq.x; a1 *= pq.x;
        a0 += bq.xy; a1 += bq.zw;
        if (pq.y == 0.f) { a0 = f32x2{0.f, 0.f}; a1 = a0; }
        if (xok && ty < rows_left) {
            __builtin_nontemporal_store(cat4(a0, a1), reinterpret_cast<f32x4*>(ob + opq(ooff)));
            if (FUSED) {
                if (s == 0) P = cat4(a0, a1);            // a thread with any pixel at all has this one
                const f32x2 d0 = a0 - P.xy, d1 = a1 - P.zw;
                va[0] += d0; va[1] += d1;
                va[2] = fma2(d0, d0, va[2]); va[3] = fma2(d1, d1, va[3]);
            }
        }
        if (more) { commit((s + 1) & 1); commit_planes(s + 1); }
        if (more2) { fetch(); fetch_planes(s + 2); }
        ob += ob_step;
        lds_barrier();
    };
    for (int s = 0; s + 2 < nsteps; ++s) step(s, true, true);
    if (nsteps > 1) step(nsteps - 2, true, false);
    step(nsteps - 1, false, false);

    if (FUSED && stats != nullptr) {
        // merge the 32 pixel lanes of every channel, re-based to a common pivot (see dw_lean.h)
        const int rows = oy_end - oy_beg, full = rows / L2_OR, tail = rows % L2_OR;
        const int cnt = xok ? full + (ty < tail ? 1 : 0) : 0;
        float* mrg = lbuf;                               // [256][13]
        static_assert(256 * 13 * 4 <= 2 * L2_BUFB, "merge buffer fits the LDS buffers");
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
            {   // common pivot: an interior lane's (lane 9 = row 1, column 1 of the step) when it saw pixels
                const float* qi = mrg + (9 * 8 + mcg) * 13;
                if (qi[0] != 0.f) { pv = qi[1 + mi]; have = true; }
            }
            for (int l = 0; l < 32; ++l) {
                const float* q = mrg + (l * 8 + mcg) * 13;
                const float n_t = q[0];
                if (n_t == 0.f) continue;
                if (!have) { pv = q[1 + mi]; have = true; }
                const float dp = q[1 + mi] - pv, b1 = q[5 + mi], b2 = q[9 + mi];
                s1 += fmaf(n_t, dp, b1);
                s2 += b2 + dp * (2.f * b1 + n_t * dp);
                nn += n_t;
            }
            const int64_t prow = (n * chunks_y + cy) * strips_x + sx;
            float* sp = stats + prow * 4 * C + (int)cb * LS_CB + ch;
            sp[0] = nn;
            sp[C] = pv;
            sp[2 * (int64_t)C] = s1;
            sp[3 * (int64_t)C] = s2;
        }
    }
}

#ifndef L2_ENABLE
#define L2_ENABLE 1              // A/B: 0 sends stride-2 forward layers back to the round-3 strip kernel
#endif
static inline bool dw_lean_s2_ok(const DtGeom& g) {
    return L2_ENABLE && g.s == 2 && g.d == 1 && !g.flip && g.c < (1 << 24) && (int64_t)(L2_ROWS * (int64_t)g.win + 64) * 4 < (1ll << 24) &&
           (int64_t)(L2_ROWS * (int64_t)g.win + 64) * g.c * 4 < (1ll << 31) &&
           (int64_t)(L2_OR * (int64_t)g.wout + 64) * g.c * 4 < (1ll << 31);
}

}  // namespace tsii
