// K2: depth-wise partial convolution (PartialConv with groups == C, same_holes;
// models/partial_convolution.py:49-80 as built at models/MobileNetV2.py:174-176).
//
// HBM-bound streaming stencil on NHWC data: one thread owns 4 consecutive channels (16-byte
// loads, a wavefront covers 256 channels = 1 KiB of one pixel per tap); the x*mask multiply
// (:51), the division by cnt*Cin (:61,71 -- the depth-wise quirk of SURVEY.md F6) and the hole
// zeroing (:72) are fused, so x is read once (taps re-hit L1/L2) and y is written once.
// The valid count comes from K1 (mask.hip) as the `denom`/`keep`/`inv` planes.
#include "tsii_common.h"

namespace tsii {

struct DwGeom {
    int n, h, w, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo;
};

// weights arrive as [C][T] (reference layout [C,1,kh,kw]); kernels read wT[T][C].
// Grids are (x: pixels-of-a-row x channel groups, y: row, z: image): all index math is 32-bit and
// the only division is by the channel-group count.
template <int W>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rmask,
                                                     const float* __restrict__ wT, const float* __restrict__ bias,
                                                     const float* __restrict__ denom, const float* __restrict__ keep,
                                                     DwGeom g, float* __restrict__ y) {
    const unsigned CG = (unsigned)(g.c / W);
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (unsigned)g.wo * CG) return;
    const int ox = (int)(j / CG);
    const int c = (int)(j % CG) * W;
    const int oy = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int64_t pix = (n * g.ho + oy) * g.wo + ox;
    VecF<W> acc;
#pragma unroll
    for (int i = 0; i < W; ++i) acc.v[i] = 0.f;
    const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;
    if (kp) {
        for (int ky = 0; ky < g.kh; ++ky) {
            const int iy = oy * g.sh - g.ph + ky * g.dh;
            if (iy < 0 || iy >= g.h) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int ix = ox * g.sw - g.pw + kx * g.dw;
                if (ix < 0 || ix >= g.w) continue;
                const int64_t ipix = (n * g.h + iy) * g.w + ix;
                const float m = rmask != nullptr ? rmask[ipix] : 1.f;
                const VecF<W> xv = vload<W>(x + ipix * g.c + c);
                const VecF<W> wv = vload<W>(wT + (ky * g.kw + kx) * g.c + c);
#pragma unroll
                for (int i = 0; i < W; ++i) acc.v[i] = fmaf(xv.v[i] * m, wv.v[i], acc.v[i]);
            }
        }
        const float dn = denom != nullptr ? denom[pix] : 1.f;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            float v = acc.v[i];
            if (denom != nullptr) v = v / dn;
            if (bias != nullptr) v += bias[c + i];
            acc.v[i] = v;
        }
    }
    vstore<W>(y + pix * g.c + c, acc);
}

template <int W>
__global__ __launch_bounds__(256) void dw_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                        const float* __restrict__ wT, const float* __restrict__ rmask,
                                                        DwGeom g, float* __restrict__ dx) {
    const unsigned CG = (unsigned)(g.c / W);
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (unsigned)g.w * CG) return;
    const int ix = (int)(j / CG);
    const int c = (int)(j % CG) * W;
    const int iy = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int64_t pix = (n * g.h + iy) * g.w + ix;
    VecF<W> acc;
#pragma unroll
    for (int i = 0; i < W; ++i) acc.v[i] = 0.f;
    const float m = rmask != nullptr ? rmask[pix] : 1.f;
    if (m != 0.f) {
        for (int ky = 0; ky < g.kh; ++ky) {
            const int ty = iy + g.ph - ky * g.dh;
            if (ty < 0 || (ty % g.sh) != 0) continue;
            const int oy = ty / g.sh;
            if (oy >= g.ho) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int tx = ix + g.pw - kx * g.dw;
                if (tx < 0 || (tx % g.sw) != 0) continue;
                const int ox = tx / g.sw;
                if (ox >= g.wo) continue;
                const int64_t opix = (n * g.ho + oy) * g.wo + ox;
                const float s = inv != nullptr ? inv[opix] : 1.f;
                const VecF<W> gv = vload<W>(dy + opix * g.c + c);
                const VecF<W> wv = vload<W>(wT + (ky * g.kw + kx) * g.c + c);
#pragma unroll
                for (int i = 0; i < W; ++i) acc.v[i] = fmaf(gv.v[i] * s, wv.v[i], acc.v[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < W; ++i) acc.v[i] *= m;
    }
    vstore<W>(dx + pix * g.c + c, acc);
}

// dW partials.  Block = CGB (<= 32) channel groups x L pixel lanes; it walks `rpb` output rows (n,oy),
// lane l taking ox = l, l+L, ...; 9 taps x W channels (+ bias) stay in registers, the L lanes are
// combined through LDS, so each block emits ONE partial row: part[by][(T+1)][C].
static constexpr int DW_TAPS = 9;
template <int W>
__global__ __launch_bounds__(256) void dw_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                        const float* __restrict__ keep, const float* __restrict__ x,
                                                        const float* __restrict__ rmask, DwGeom g, int CGB, int L,
                                                        int rpb, float* __restrict__ part) {
    __shared__ float red[256];
    const int CG = g.c / W;
    const int T = g.kh * g.kw;
    const int cgl = threadIdx.x % CGB, lane = threadIdx.x / CGB;
    const int cg = blockIdx.x * CGB + cgl;
    const bool active = (lane < L) && (cg < CG);
    const int c = cg * W;
    const int rows_total = g.n * g.ho;
    const int row0 = blockIdx.y * rpb;
    const int row1 = row0 + rpb < rows_total ? row0 + rpb : rows_total;
    float* prow = part + (int64_t)blockIdx.y * (T + 1) * g.c;
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
                    if (keep != nullptr && keep[pix] == 0.f) continue;  // hole: no gradient (partial_convolution.py:72)
                    const float s = inv != nullptr ? inv[pix] : 1.f;
                    VecF<W> gv = vload<W>(dy + pix * g.c + c);
#pragma unroll
                    for (int i = 0; i < W; ++i) { acc[DW_TAPS][i] += gv.v[i]; gv.v[i] *= s; }  // bias is added after the division
#pragma unroll
                    for (int t = 0; t < DW_TAPS; ++t) {
                        const int tt = t0 + t;
                        if (tt >= T) break;
                        const int ky = tt / g.kw, kx = tt % g.kw;
                        const int iy = oy * g.sh - g.ph + ky * g.dh;
                        const int ix = ox * g.sw - g.pw + kx * g.dw;
                        if (iy < 0 || iy >= g.h || ix < 0 || ix >= g.w) continue;
                        const int64_t ipix = (n * g.h + iy) * g.w + ix;
                        const float m = rmask != nullptr ? rmask[ipix] : 1.f;
                        if (m == 0.f) continue;
                        const VecF<W> xv = vload<W>(x + ipix * g.c + c);
#pragma unroll
                        for (int i = 0; i < W; ++i) acc[t][i] = fmaf(gv.v[i], xv.v[i] * m, acc[t][i]);
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

// sum the R partial rows (block = 32 columns x 8 row lanes, 4 loads in flight) and scatter back to the
// reference layout dw[c][t], db[c]
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ part, int R, int T, int C,
                                                        float* __restrict__ dwgt, float* __restrict__ dbias) {
    __shared__ double sh[8][33];
    const int64_t len = (int64_t)(T + 1) * C;
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

extern "C" int tsii_dw_fwd(const float* x, const float* rmask, const float* w, const float* bias,
                           const float* denom, const float* keep, int n, int h, int wd, int c, int kh, int kw,
                           int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo, float* y, float* ws,
                           void* stream) {
    TSII_REQUIRE(x && w && y && ws, "dw_fwd: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_fwd")) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_transpose(w, c, kh * kw, ws, st);  // [C][T] -> [T][C]
    if (rc) return rc;
    const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(ws);
    TSII_REQUIRE(ho <= 65535 && n <= 65535, "dw_fwd: grid limit");
    const dim3 grid(cdiv(wo * (vec ? c / 4 : c), 256), ho, n);
    if (vec) hipLaunchKernelGGL((dw_fwd_kernel<4>), grid, dim3(256), 0, st, x, rmask, ws, bias, denom, keep, g, y);
    else hipLaunchKernelGGL((dw_fwd_kernel<1>), grid, dim3(256), 0, st, x, rmask, ws, bias, denom, keep, g, y);
    return check_launch("dw_fwd");
}

extern "C" int tsii_dw_bwd_dx(const float* dy, const float* inv, const float* w, const float* rmask,
                              int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* dx, float* ws, void* stream) {
    TSII_REQUIRE(dy && w && dx && ws, "dw_bwd_dx: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_bwd_dx")) return -1;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_transpose(w, c, kh * kw, ws, st);
    if (rc) return rc;
    const bool vec = (c % 4 == 0) && aligned16(dy) && aligned16(dx) && aligned16(ws);
    TSII_REQUIRE(h <= 65535 && n <= 65535, "dw_bwd_dx: grid limit");
    const dim3 grid(cdiv(wd * (vec ? c / 4 : c), 256), h, n);
    if (vec) hipLaunchKernelGGL((dw_bwd_dx_kernel<4>), grid, dim3(256), 0, st, dy, inv, ws, rmask, g, dx);
    else hipLaunchKernelGGL((dw_bwd_dx_kernel<1>), grid, dim3(256), 0, st, dy, inv, ws, rmask, g, dx);
    return check_launch("dw_bwd_dx");
}

extern "C" size_t tsii_dw_bwd_dw_ws_bytes(int n, int ho, int wo, int c, int kh, int kw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || kh <= 0 || kw <= 0) return 0;
    // the scalar plan (taken when c % 4 != 0 or a pointer is unaligned) never needs more rows
    const DwPlan a = plan_dw(n, ho, c, c % 4 == 0), b = plan_dw(n, ho, c, false);
    const int R = a.R > b.R ? a.R : b.R;
    return (size_t)R * (size_t)(kh * kw + 1) * c * sizeof(float);
}

extern "C" int tsii_dw_bwd_dw(const float* dy, const float* inv, const float* keep, const float* x, const float* rmask,
                              int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* dwgt, float* dbias, void* ws, size_t ws_bytes,
                              void* stream) {
    TSII_REQUIRE(dy && x && dwgt && ws, "dw_bwd_dw: null pointer");
    DW_GEOM();
    if (check_geom(g, "dw_bwd_dw")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dw_bwd_dw_ws_bytes(n, ho, wo, c, kh, kw), "dw_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && aligned16(dy) && aligned16(x);
    const DwPlan p = plan_dw(n, ho, c, vec);
    float* part = (float*)ws;
    const dim3 grid(p.gx, p.gy);
    if (vec) hipLaunchKernelGGL((dw_bwd_dw_kernel<4>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, g, p.CGB, p.L, p.rpb, part);
    else hipLaunchKernelGGL((dw_bwd_dw_kernel<1>), grid, dim3(256), 0, st, dy, inv, keep, x, rmask, g, p.CGB, p.L, p.rpb, part);
    int rc = check_launch("dw_bwd_dw");
    if (rc) return rc;
    const int T = kh * kw;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((unsigned)cdiv64((int64_t)(T + 1) * c, 32)), dim3(256), 0, st, part, p.R, T, c, dwgt, dbias);
    return check_launch("dw_reduce");
}
