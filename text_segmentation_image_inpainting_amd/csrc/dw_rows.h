// K2 (round 6): 3x3 depth-wise stencils with LARGE dilation on mid-sized maps -- one ROW PHASE of a channel block in LDS.
//
// TextSegament's dilated MobileNetV2 stages and its RFB run 3x3 depth-wise convolutions with dilation 8 / 16 (17, 29) on 64 x 64
// maps (models/MobileNetV2.py:114-149, models/common.py:96-156; 1920 channels after expansion at 512^2).  The marching-strip kernels
// stage a ring of 8 + 2 d rows x (16 + 2 d) columns per 16-pixel strip -- 98 KB of LDS, one block per CU, every input read twice in
// x: 1.7 - 3.1 TB/s at dilation 8 (cfg 3: 16 ms per step) -- and dilation 16 has no strip form at all (direct kernels, 2 TB/s).
// dw_small.h (the whole map in LDS) ends at 1024 pixels.
//
// With dilation d the rows y = p (mod d) of an image only ever meet each other: the outputs of row phase p read inputs of row
// phase p (for "same" padding d), at rows one apart IN THE PHASE and columns d apart in the row.  A phase of a 64 x 64 map at
// d = 8 is 8 rows x 64 pixels = 64 KB for a 32-channel block: it fits LDS whole, so there is no halo at all -- every input is read
// once, every output written once.  A block owns (image, 32 channels) and walks the d phases with TWO phase buffers: the loads of
// phase p + 1 are issued before phase p is computed and committed after it, one LDS barrier per phase.  A thread owns one column
// of the phase and 4 channels: per tap column 8 reads feed 24 float2 FMAs x 2 (3 tap rows x 8 outputs).
//   MODE 0 plain | 1 forward with BatchNorm on load and / or statistics partials (K6b) | 2 dX feeding a BatchNorm backward (K6c)
// No mask planes (the callers with planes keep the strip kernels); partial rows keep the strip plan's layout like dw_small.h: an
// image's first row carries its sums, its other rows are written empty.
// Included by dwconv.hip after dw_small.h (f32x2 helpers of dw_lean.h).
#pragma once

namespace tsii {

static constexpr int DR_CB = 32;                  // channels per block (one 128-byte segment per pixel)
static constexpr int DR_ROWS = 8;                 // rows of a phase
static constexpr int DR_W = 64;                   // columns
static constexpr int DR_THREADS = 512;            // 8 channel quads x 64 columns
static constexpr int DR_NLD = DR_ROWS * DR_W * 8 / DR_THREADS;     // 16-byte items per thread and phase (8)
static constexpr int DR_BUFB = DR_ROWS * DR_W * DR_CB * 4;          // bytes of a phase buffer (65536)

#ifdef TSII_HIP_EMU
static int g_dr_enable = 1;      // TEST-ONLY: 0 keeps these geometries on the strip / direct kernels (the CPU suite runs both)
extern "C" void tsii_emu_set_dw_rows(int v) { g_dr_enable = v; }
#define DR_ENABLE g_dr_enable
#else
#ifndef DR_ENABLE
#define DR_ENABLE 1              // A/B: 0 sends these geometries back to the strip / direct kernels
#endif
#endif
#ifndef DR_MIN_DIL
#define DR_MIN_DIL 8
#endif

// a function of the grid the kernel writes (= reads: "same" convolution) alone, like every choice the partial-row layout hangs on
static inline bool dw_rows_dims_ok(int h, int w, int c, int s, int d) {
    return DR_ENABLE && s == 1 && d >= DR_MIN_DIL && c % 4 == 0 && w <= DR_W && w >= 8 && (h + d - 1) / d <= DR_ROWS && h >= d &&
           (int64_t)h * w > SM_MAXPIX && (int64_t)h * w * c * 4 < (1ll << 31);
}
static inline bool dw_rows_ok(const DtGeom& g) {
    return dw_rows_dims_ok(g.hout, g.wout, g.c, g.s, g.d) && g.hin == g.hout && g.win == g.wout && g.pad_h == g.d && g.pad_w == g.d;
}

template <int MODE>
__global__ __launch_bounds__(DR_THREADS, 1) void dw_rows_kernel(const float* __restrict__ in, const float* __restrict__ wT, const float* __restrict__ bias,
                                                                DtGeom g, unsigned cblocks, unsigned rows_per_image, DwBN ib, float* __restrict__ stats,
                                                                DwBnBwd bb, float* __restrict__ out) {
    constexpr bool FUSED = (MODE == 1), BNB = (MODE == 2);
    __shared__ __attribute__((aligned(16))) unsigned char lbuf[2 * DR_BUFB];
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned cb = b % cblocks;
    const int64_t n = b / cblocks;
    const int t = threadIdx.x, cg = t & 7, x = t >> 3;            // this thread's channel quad and output column
    const int C = g.c, W = g.win, H = g.hin, D = g.d;
    const int c0 = (int)cb * DR_CB + cg * 4;
    const bool cok = c0 < C, xok = cok && x < W;
    const unsigned c0b = (unsigned)(cok ? c0 : C - 4) * 4u;
    const bool bn_in = FUSED && ib.sc != nullptr;
    const bool hi_finite = bn_in && ib.hi < __builtin_huge_valf();
    const float bn_neg = bn_in ? ib.neg : 1.f;

    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = cok ? *reinterpret_cast<const f32x4*>(wT + (g.flip ? 8 - k : k) * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f}, bq = {0.f, 0.f, 0.f, 0.f};
    if (bn_in && cok) { isc = *reinterpret_cast<const f32x4*>(ib.sc + c0); ish = *reinterpret_cast<const f32x4*>(ib.sh + c0); }
    if (!BNB && bias != nullptr && cok) bq = *reinterpret_cast<const f32x4*>(bias + c0);
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bga = bmu, bbe = bmu;
    if (BNB && cok) {
        bmu = *reinterpret_cast<const f32x4*>(bb.mean + c0);
        const f32x4 var = *reinterpret_cast<const f32x4*>(bb.var + c0);
        bis = f32x4{1.0f / sqrtf(var.x + bb.eps), 1.0f / sqrtf(var.y + bb.eps), 1.0f / sqrtf(var.z + bb.eps), 1.0f / sqrtf(var.w + bb.eps)};
        bga = *reinterpret_cast<const f32x4*>(bb.gamma + c0);
        bbe = *reinterpret_cast<const f32x4*>(bb.beta + c0);
    }

    // items of a phase: 16-byte item q = t + 512 i is channel quad q & 7 of phase pixel q >> 3 = (row, column); LDS byte q * 16.
    // Item i's pixel is 64 i pixels after the thread's first one: its (row, column) are recomputed where they are used (a handful
    // of integer operations per phase) instead of living in 16 registers.
    const int p0 = t >> 3, r0 = p0 / W, x0 = p0 - r0 * W, rstep = DR_W / W, xstep = DR_W - rstep * W;      // 64 = rstep W + xstep
    auto item_rc = [&](int i, int& r, int& px) {
        px = x0 + i * xstep; r = r0 + i * rstep;
        while (px >= W) { px -= W; ++r; }
    };
    const char* const ibase = reinterpret_cast<const char*>(in + n * (int64_t)H * W * C);
    char* const obase = reinterpret_cast<char*>(out + n * (int64_t)H * W * C);
    const char* const ybase = BNB ? reinterpret_cast<const char*>(bb.y + n * (int64_t)H * W * C) : nullptr;
    const unsigned rowb = (unsigned)W * (unsigned)C * 4u;        // bytes of an image row
    auto rows_of = [&](int py) { return (H - py + D - 1) / D; };  // rows of phase py (H >= d: at least one)
    f32x4 pf[DR_NLD];
    auto fetch = [&](int py) {
        const int R = rows_of(py);
        const char* const pb = ibase + (int64_t)py * rowb + c0b;
#pragma unroll
        for (int i = 0; i < DR_NLD; ++i) {
            int r, px;
            item_rc(i, r, px);
            // rows past the phase's last re-read its last row (a valid address); they are committed as zeros
            const int rc = r < R ? r : R - 1;
            pf[i] = *reinterpret_cast<const f32x4*>(pb + (unsigned)rc * (unsigned)D * rowb + (unsigned)px * (unsigned)C * 4u);
        }
    };
    auto commit = [&](int which, int py) {
        const int R = rows_of(py);
        unsigned char* const T = lbuf + which * DR_BUFB + t * 16;
#pragma unroll
        for (int i = 0; i < DR_NLD; ++i) {
            int r, px;
            item_rc(i, r, px);
            f32x4 v = pf[i];
            if (FUSED) {
                f32x2 z0 = fma2(v.xy, isc.xy, ish.xy), z1 = fma2(v.zw, isc.zw, ish.zw);
                z0 = max2(z0, z0 * bn_neg); z1 = max2(z1, z1 * bn_neg);
                if (hi_finite) { z0 = min2(z0, f32x2{ib.hi, ib.hi}); z1 = min2(z1, f32x2{ib.hi, ib.hi}); }
                v = cat4(z0, z1);
            }
            if (r >= R) v = f32x4{0.f, 0.f, 0.f, 0.f};                 // below the phase's last row: zero padding
            *reinterpret_cast<f32x4*>(T + DR_THREADS * 16 * i) = v;
        }
    };

    f32x4 P = {0.f, 0.f, 0.f, 0.f};                      // K6b: thread-local pivot = its first output
    bool have_p = false;
    int cnt = 0;
    f32x2 va[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // K6b: s1.xy s1.zw s2.xy s2.zw | K6c: sum dz, sum dz*xhat

    const int nph = D < H ? D : H;
    fetch(0);
    commit(0, 0);
    if (nph > 1) fetch(1);
    __syncthreads();
    for (int py = 0; py < nph; ++py) {
        const int R = rows_of(py);
        const unsigned char* const S = lbuf + (py & 1) * DR_BUFB + cg * 16;
        // Rows in groups of GR: all 8 at once (24 reads per phase) -- or, for K6c, two groups of 4 (36 reads): its 16 BatchNorm
        // constants, the BatchNorm-input rows and the reductions do not fit beside 8 rows of accumulators and tap values (the
        // 8-row form kept 33 loop-invariant registers in scratch and reloaded them every phase: 2.09 ms where the forward takes 0.83)
        constexpr int GR = BNB ? 4 : DR_ROWS;
        char* const ob = obase + (int64_t)py * rowb + (unsigned)x * (unsigned)C * 4u + (unsigned)c0 * 4u;
        const char* const yb = BNB ? ybase + (int64_t)py * rowb + (unsigned)(x < W ? x : 0) * (unsigned)C * 4u + c0b : nullptr;
#pragma unroll
        for (int rb = 0; rb < DR_ROWS; rb += GR) {
            if (rb >= R) break;                                        // block-uniform
            // K6c: the raw BatchNorm input at this group's outputs, requested BEFORE the taps (in flight across the multiply-adds;
            // rows past the phase's last re-read its last row)
            f32x4 yv[BNB ? GR : 1];
            if (BNB) {
#pragma unroll
                for (int r = 0; r < GR; ++r) yv[r] = *reinterpret_cast<const f32x4*>(yb + (unsigned)(rb + r < R ? rb + r : R - 1) * (unsigned)D * rowb);
            }
            f32x2 a[GR][2];
#pragma unroll
            for (int r = 0; r < GR; ++r) { a[r][0] = f32x2{0.f, 0.f}; a[r][1] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xc = x + (kx - 1) * D;
                const bool cin = xc >= 0 && xc < W;
                const int xcc = cin ? xc : x;
                f32x4 v[GR + 2];                                       // phase rows rb - 1 .. rb + GR
#pragma unroll
                for (int j = 0; j < GR + 2; ++j) {
                    const int ri = rb + j - 1;
                    if (ri < 0 || ri >= DR_ROWS) { v[j] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }      // above / below the phase: zero padding
                    v[j] = *reinterpret_cast<const f32x4*>(S + (ri * W + (x < W ? xcc : 0)) * (DR_CB * 4));
                    if (!cin) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};        // a tap column outside the image: zero padding
                }
#pragma unroll
                for (int r = 0; r < GR; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {                   // (rows R .. 7 of the buffer hold zeros)
                        a[r][0] = fma2(v[r + ky].xy, w[ky * 3 + kx].xy, a[r][0]);
                        a[r][1] = fma2(v[r + ky].zw, w[ky * 3 + kx].zw, a[r][1]);
                    }
            }
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                if (rb + r >= R) break;                                // block-uniform
                f32x2 a0 = a[r][0], a1 = a[r][1];
                if (!BNB) { a0 += bq.xy; a1 += bq.zw; }
                if (xok) {
                    __builtin_nontemporal_store(cat4(a0, a1), reinterpret_cast<f32x4*>(ob + (unsigned)(rb + r) * (unsigned)D * rowb));
                    if (FUSED) {
                        if (!have_p) { P = cat4(a0, a1); have_p = true; }
                        const f32x2 d0 = a0 - P.xy, d1 = a1 - P.zw;
                        va[0] += d0; va[1] += d1;
                        va[2] = fma2(d0, d0, va[2]); va[3] = fma2(d1, d1, va[3]);
                        ++cnt;
                    }
                }
                if (BNB) {
                    const f32x4 yq = yv[r];
                    const f32x2 h0 = (yq.xy - bmu.xy) * bis.xy, h1 = (yq.zw - bmu.zw) * bis.zw;
                    const f32x2 z0 = fma2(h0, bga.xy, bbe.xy), z1 = fma2(h1, bga.zw, bbe.zw);
                    f32x2 d0 = a0, d1 = a1;
                    d0.x *= (z0.x > 0.f && z0.x < bb.hi) ? 1.f : (z0.x > 0.f ? 0.f : bb.neg);
                    d0.y *= (z0.y > 0.f && z0.y < bb.hi) ? 1.f : (z0.y > 0.f ? 0.f : bb.neg);
                    d1.x *= (z1.x > 0.f && z1.x < bb.hi) ? 1.f : (z1.x > 0.f ? 0.f : bb.neg);
                    d1.y *= (z1.y > 0.f && z1.y < bb.hi) ? 1.f : (z1.y > 0.f ? 0.f : bb.neg);
                    if (xok) {
                        va[0] += d0; va[1] += d1;
                        va[2] = fma2(d0, h0, va[2]); va[3] = fma2(d1, h1, va[3]);
                    }
                }
            }
        }
        if (py + 1 < nph) commit((py + 1) & 1, py + 1);                // into the buffer nobody reads during this phase
        if (py + 2 < nph) fetch(py + 2);
        lds_barrier();
    }

    // ---- partial rows: the image's first row carries the block's sums, its other rows are empty ------------------------------
    if (BNB || (FUSED && stats != nullptr)) {
        float* const mrg = reinterpret_cast<float*>(lbuf);            // [512][13]; the loop ended on a barrier
        float* const mt = mrg + t * 13;
        mt[0] = (float)cnt;
        mt[1] = P.x; mt[2] = P.y; mt[3] = P.z; mt[4] = P.w;
        mt[5] = va[0].x; mt[6] = va[0].y; mt[7] = va[1].x; mt[8] = va[1].y;
        mt[9] = va[2].x; mt[10] = va[2].y; mt[11] = va[3].x; mt[12] = va[3].y;
        __syncthreads();
        if (t < DR_CB && (int)cb * DR_CB + t < C) {
            const int ch = t, mcg = ch / 4, mi = ch % 4;
            const int64_t prow = n * rows_per_image;
            if (BNB) {
                float s1 = 0.f, s2 = 0.f;
                for (int l = 0; l < 64; ++l) { const float* q = mrg + (l * 8 + mcg) * 13; s1 += q[5 + mi]; s2 += q[9 + mi]; }
                float* sp = bb.part + prow * 2 * C + (int)cb * DR_CB + ch;
                sp[0] = s1;
                sp[C] = s2;
                for (unsigned r = 1; r < rows_per_image; ++r) { sp[(int64_t)r * 2 * C] = 0.f; sp[(int64_t)r * 2 * C + C] = 0.f; }
            } else {
                // (count, pivot, s1, s2) per thread, re-based to a common pivot:  s1' = s1 + n dp,  s2' = s2 + 2 dp s1 + n dp^2
                float nn = 0.f, pv = 0.f, s1 = 0.f, s2 = 0.f;
                bool have = false;
                for (int l = 0; l < 64; ++l) {
                    const float* q = mrg + (l * 8 + mcg) * 13;
                    const float n_t = q[0];
                    if (n_t == 0.f) continue;
                    if (!have) { pv = q[1 + mi]; have = true; }
                    const float dp = q[1 + mi] - pv, a1 = q[5 + mi], a2 = q[9 + mi];
                    s1 += fmaf(n_t, dp, a1);
                    s2 += a2 + dp * (2.f * a1 + n_t * dp);
                    nn += n_t;
                }
                float* sp = stats + prow * 4 * C + (int)cb * DR_CB + ch;
                sp[0] = nn;
                sp[C] = pv;
                sp[2 * (int64_t)C] = s1;
                sp[3 * (int64_t)C] = s2;
                for (unsigned r = 1; r < rows_per_image; ++r) {
                    float* sr = sp + (int64_t)r * 4 * C;
                    sr[0] = 0.f; sr[C] = 0.f; sr[2 * (int64_t)C] = 0.f; sr[3 * (int64_t)C] = 0.f;
                }
            }
        }
    }
}

// ---- weight gradient: dw[c][ky][kx] = sum dy[n, y, x, c] * a[n, y + (ky - 1) d, x + (kx - 1) d, c], bias gradient = sum dy -------
// Same staging (a = the input, with the producer's BatchNorm + activation applied on load when given); dy straight from memory at
// the thread's output positions; 9 taps x 4 channels (+ the bias column) accumulate in registers over all phases of the image; one
// partial row [10][C] per image, summed by dw_reduce_kernel.
template <bool BNIN>
__global__ __launch_bounds__(DR_THREADS, 1) void dw_rows_dw_kernel(const float* __restrict__ dy, const float* __restrict__ xin, DtGeom g, unsigned cblocks,
                                                                   DwBN ib, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned char lbuf[2 * DR_BUFB];
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned cb = b % cblocks;
    const int64_t n = b / cblocks;
    const int t = threadIdx.x, cg = t & 7, x = t >> 3;
    const int C = g.c, W = g.win, H = g.hin, D = g.d;
    const int c0 = (int)cb * DR_CB + cg * 4;
    const bool cok = c0 < C, xok = cok && x < W;
    const unsigned c0b = (unsigned)(cok ? c0 : C - 4) * 4u;
    const bool hi_finite = BNIN && ib.hi < __builtin_huge_valf();
    f32x4 isc = {1.f, 1.f, 1.f, 1.f}, ish = {0.f, 0.f, 0.f, 0.f};
    if (BNIN && cok) { isc = *reinterpret_cast<const f32x4*>(ib.sc + c0); ish = *reinterpret_cast<const f32x4*>(ib.sh + c0); }
    const int p0 = t >> 3, r0 = p0 / W, x0 = p0 - r0 * W, rstep = DR_W / W, xstep = DR_W - rstep * W;      // (see dw_rows_kernel)
    auto item_rc = [&](int i, int& r, int& px) {
        px = x0 + i * xstep; r = r0 + i * rstep;
        while (px >= W) { px -= W; ++r; }
    };
    const char* const ibase = reinterpret_cast<const char*>(xin + n * (int64_t)H * W * C);
    const char* const gbase = reinterpret_cast<const char*>(dy + n * (int64_t)H * W * C);
    const unsigned rowb = (unsigned)W * (unsigned)C * 4u;
    auto rows_of = [&](int py) { return (H - py + D - 1) / D; };
    f32x4 pf[DR_NLD];
    auto fetch = [&](int py) {
        const int R = rows_of(py);
        const char* const pb = ibase + (int64_t)py * rowb + c0b;
#pragma unroll
        for (int i = 0; i < DR_NLD; ++i) {
            int r, px;
            item_rc(i, r, px);
            const int rc = r < R ? r : R - 1;
            pf[i] = *reinterpret_cast<const f32x4*>(pb + (unsigned)rc * (unsigned)D * rowb + (unsigned)px * (unsigned)C * 4u);
        }
    };
    auto commit = [&](int which, int py) {
        const int R = rows_of(py);
        unsigned char* const T = lbuf + which * DR_BUFB + t * 16;
#pragma unroll
        for (int i = 0; i < DR_NLD; ++i) {
            int r, px;
            item_rc(i, r, px);
            f32x4 v = pf[i];
            if (BNIN) {
                f32x2 z0 = fma2(v.xy, isc.xy, ish.xy), z1 = fma2(v.zw, isc.zw, ish.zw);
                z0 = max2(z0, z0 * ib.neg); z1 = max2(z1, z1 * ib.neg);
                if (hi_finite) { z0 = min2(z0, f32x2{ib.hi, ib.hi}); z1 = min2(z1, f32x2{ib.hi, ib.hi}); }
                v = cat4(z0, z1);
            }
            if (r >= R) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(T + DR_THREADS * 16 * i) = v;
        }
    };
    f32x2 acc[10][2];
#pragma unroll
    for (int k = 0; k < 10; ++k) { acc[k][0] = f32x2{0.f, 0.f}; acc[k][1] = f32x2{0.f, 0.f}; }
    const int nph = D < H ? D : H;
    fetch(0);
    commit(0, 0);
    if (nph > 1) fetch(1);
    __syncthreads();
    for (int py = 0; py < nph; ++py) {
        const int R = rows_of(py);
        const unsigned char* const S = lbuf + (py & 1) * DR_BUFB + cg * 16;
        // rows in groups of GR (all 8; 4 with BatchNorm on load, whose constants do not fit beside 8 rows of dy and tap values): this
        // thread's dy rows of the group (rows past the phase's last and columns past the image: zeros), then the 3 tap columns
        constexpr int GR = BNIN ? 4 : DR_ROWS;
        const char* const gb = gbase + (int64_t)py * rowb + (unsigned)(x < W ? x : 0) * (unsigned)C * 4u + c0b;
#pragma unroll
        for (int rb = 0; rb < DR_ROWS; rb += GR) {
            if (rb >= R) break;                                        // block-uniform
            f32x4 gq[GR];
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                gq[r] = *reinterpret_cast<const f32x4*>(gb + (unsigned)(rb + r < R ? rb + r : R - 1) * (unsigned)D * rowb);
                if (rb + r >= R || !xok) gq[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[9][0] += gq[r].xy; acc[9][1] += gq[r].zw;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xc = x + (kx - 1) * D;
                const bool cin = xc >= 0 && xc < W;
                const int xcc = cin ? xc : x;
                f32x4 v[GR + 2];                                       // phase rows rb - 1 .. rb + GR
#pragma unroll
                for (int j = 0; j < GR + 2; ++j) {
                    const int ri = rb + j - 1;
                    if (ri < 0 || ri >= DR_ROWS) { v[j] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                    v[j] = *reinterpret_cast<const f32x4*>(S + (ri * W + (x < W ? xcc : 0)) * (DR_CB * 4));
                    if (!cin) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int r = 0; r < GR; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        acc[ky * 3 + kx][0] = fma2(gq[r].xy, v[r + ky].xy, acc[ky * 3 + kx][0]);
                        acc[ky * 3 + kx][1] = fma2(gq[r].zw, v[r + ky].zw, acc[ky * 3 + kx][1]);
                    }
            }
        }
        if (py + 1 < nph) commit((py + 1) & 1, py + 1);
        if (py + 2 < nph) fetch(py + 2);
        lds_barrier();
    }
    // combine the 64 column lanes of every channel: [10][512] float4 through the (free) buffers, 5 taps at a time
    f32x4* const red4 = reinterpret_cast<f32x4*>(lbuf);
    float* const prow = part + n * 10 * (int64_t)C;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; ++k) red4[k * DR_THREADS + t] = cat4(acc[half * 5 + k][0], acc[half * 5 + k][1]);
        __syncthreads();
        if (t < 5 * DR_CB) {
            const int k = t / DR_CB, ch = t % DR_CB;
            if ((int)cb * DR_CB + ch < C) {
                const float* col = reinterpret_cast<const float*>(lbuf) + (k * DR_THREADS) * 4 + (ch / 4) * 4 + (ch % 4);
                float sum = 0.f;
#pragma unroll 8
                for (int l = 0; l < 64; ++l) sum += col[l * 8 * 4];
                prow[(int64_t)(half * 5 + k) * C + (int)cb * DR_CB + ch] = sum;
            }
        }
    }
}

}  // namespace tsii
