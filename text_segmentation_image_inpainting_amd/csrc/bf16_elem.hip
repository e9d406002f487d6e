// bf16 ACTIVATION STORAGE, streaming kernels of the segmentation nets: BatchNorm statistics / apply / backward
// (nn.Sequential(BatchNorm2d, act), models/BaseModels.py:95-99), residual add (models/Xception.py:44, models/MobileNetV2.py:146-147),
// channel concat (models/text_segmentation.py:68,75,80,111; models/common.py:91), bilinear up-sampling
// (models/text_segmentation.py:54,76,109,113), the 3-channel stem's space-to-depth rearrangement and the casts at the
// fp32 <-> bf16 boundary (network input, logits).  All HBM-bound: ONE 16-byte vector (8 bf16) per thread on a flat grid with
// non-temporal loads / stores (tsii_common.h: flat_grid), fp32 arithmetic, one RNE rounding per stored value.
#include "bf16_common.h"

namespace tsii {

// ---- BatchNorm statistics: partial rows in the layout of the producer-side fusion (K6b): [rows][4][c] = (count, pivot, s1, s2) --
__global__ __launch_bounds__(256) void hbn_stats_kernel(const bf16_t* __restrict__ y, int64_t M, int C, int R, float* __restrict__ part) {
    const int G = C / 8;
    const int64_t tasks = (int64_t)R * G;
    for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < tasks; task += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(task % G) * 8;
        const int r = (int)(task / G);
        float pv[8], s1[8], s2[8];
        unpack8(ld8(y + (int64_t)r * C + c), pv);          // pivot: the partial row's own first value (r < R <= M)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        int64_t cnt = 0;
        for (int64_t m = r; m < M; m += R, ++cnt) {
            float v[8];
            unpack8(ld8(y + m * C + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[e] - pv[e];
                s1[e] += d;
                s2[e] = fmaf(d, d, s2[e]);
            }
        }
        float* p = part + (int64_t)r * 4 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) { p[c + e] = (float)cnt; p[C + c + e] = pv[e]; p[2 * C + c + e] = s1[e]; p[3 * C + c + e] = s2[e]; }
    }
}

// out = act(scale * y + shift) (+ residual): exactly what a load-time consumer of the same BatchNorm computes (K6b)
__global__ __launch_bounds__(256) void hbn_apply_kernel(const bf16_t* __restrict__ y, int64_t M, int C, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act, float slope, const bf16_t* __restrict__ residual,
                                                        bf16_t* __restrict__ out) {
    const unsigned G = (unsigned)(C / 8);
    const int64_t total = M * G;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % G) * 8;
    const VecF<4> s0 = vload<4>(scale + c), s1 = vload<4>(scale + c + 4), h0 = vload<4>(shift + c), h1 = vload<4>(shift + c + 4);
    const float sc[8] = {s0.v[0], s0.v[1], s0.v[2], s0.v[3], s1.v[0], s1.v[1], s1.v[2], s1.v[3]};
    const float sh[8] = {h0.v[0], h0.v[1], h0.v[2], h0.v[3], h1.v[0], h1.v[1], h1.v[2], h1.v[3]};
    float v[8], rv[8];
    unpack8(ld8_nt(y + gt * 8), v);
    if (residual != nullptr) unpack8(ld8_nt(residual + gt * 8), rv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float z = apply_act(fmaf(v[e], sc[e], sh[e]), act, slope);
        if (residual != nullptr) z += rv[e];
        v[e] = z;
    }
    st8_nt(out + gt * 8, pack8(v));
}

// backward pass 1: per-channel s1 = sum dz, s2 = sum dz * xhat, dz = dout * act'(z); partial rows [R][2][C]
__global__ __launch_bounds__(256) void hbn_bwd_partial_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y, int64_t M, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ var,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              int act, float slope, int R, float* __restrict__ part) {
    const int G = C / 8;
    const int64_t tasks = (int64_t)R * G;
    for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < tasks; task += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(task % G) * 8;
        const int r = (int)(task / G);
        float mu[8], istd[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mu[e] = mean[c + e]; istd[e] = 1.0f / sqrtf(var[c + e] + eps);
            ga[e] = gamma[c + e]; be[e] = beta[c + e]; s1[e] = 0.f; s2[e] = 0.f;
        }
        for (int64_t m = r; m < M; m += R) {
            float yv[8], dv[8];
            unpack8(ld8(y + m * C + c), yv);
            unpack8(ld8(dout + m * C + c), dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (yv[e] - mu[e]) * istd[e];
                const float z = fmaf(xh, ga[e], be[e]);
                const float dz = dv[e] * act_grad(z, act, slope);
                s1[e] += dz;
                s2[e] = fmaf(dz, xh, s2[e]);
            }
        }
        float* p = part + (int64_t)r * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) { p[c + e] = s1[e]; p[C + c + e] = s2[e]; }
    }
}

// backward pass 2: dy = gamma * istd * (dz - s1/M - xhat * s2/M)  (training)  |  gamma * istd * dz  (eval)
// A thread owns ONE 8-channel group and RPT consecutive rows: its 48 per-channel constants (192 bytes through the vector cache) are
// fetched once per RPT x 48 bytes of streamed data -- with one row per thread the constants were four times the streamed bytes and
// the pass ran at 4.3 TB/s where the forward apply (two constants per channel) runs at 6.7.
template <int RPT>
__global__ __launch_bounds__(256) void hbn_bwd_apply_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y, int64_t M, int C,
                                                            const float* __restrict__ coef, int act, float slope, bf16_t* __restrict__ dy) {
    const unsigned G = (unsigned)(C / 8);
    const int64_t total = ((M + RPT - 1) / RPT) * G;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % G) * 8;
    const int64_t r0 = (gt / G) * RPT;
    float k[6][8];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const VecF<4> a = vload<4>(coef + j * C + c), b = vload<4>(coef + j * C + c + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { k[j][i] = a.v[i]; k[j][4 + i] = b.v[i]; }
    }
    hu32x4 yq[RPT], dq[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {              // every load of the thread in flight before the first use
        const int64_t row = r0 + i < M ? r0 + i : M - 1;
        yq[i] = ld8_nt(y + row * C + c);
        dq[i] = ld8_nt(dout + row * C + c);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        if (r0 + i >= M) break;
        float yv[8], dv[8];
        unpack8(yq[i], yv);
        unpack8(dq[i], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = (yv[e] - k[0][e]) * k[1][e];
            const float z = fmaf(xh, k[2][e], k[3][e]);
            float dz = dv[e] * act_grad(z, act, slope);
            dz = dz - k[4][e] - xh * k[5][e];
            dv[e] = dz * k[2][e] * k[1][e];
        }
        st8_nt(dy + (r0 + i) * C + c, pack8(dv));
    }
}

// ---- residual add (+ activation), activation backward ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hadd_act_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, int64_t n8, int act, float slope,
                                                       bf16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float u[8], v[8];
    unpack8(ld8_nt(a + i * 8), u);
    unpack8(ld8_nt(b + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = apply_act(u[e] + v[e], act, slope);
    st8_nt(out + i * 8, pack8(u));
}
__global__ __launch_bounds__(256) void hact_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, int64_t n8, int act, float slope,
                                                       bf16_t* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float d[8], v[8];
    unpack8(ld8_nt(dout + i * 8), d);
    unpack8(ld8_nt(x + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= act_grad(v[e], act, slope);
    st8_nt(dx + i * 8, pack8(d));
}

// ---- channel concat / slice on [M, C] rows -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hcopy_channels_kernel(bf16_t* __restrict__ big, int64_t m, int cbig, int coff, bf16_t* __restrict__ sm_,
                                                             int csm, int to_dst) {
    const unsigned G = (unsigned)(csm / 8);
    const int64_t total = m * G;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % G) * 8;
    const int64_t row = gt / G;
    bf16_t* pb = big + row * cbig + coff + c;
    bf16_t* ps = sm_ + row * csm + c;
    if (to_dst) st8(pb, ld8_nt(ps));
    else st8_nt(ps, ld8(pb));
}

// ---- bilinear up-sampling by an integer factor, align_corners = False ----------------------------------------------------------------
__device__ __forceinline__ void hbilin_src(int o, int scale, int limit, int& i0, int& i1, float& l1) {
    float s = ((float)o + 0.5f) / (float)scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > limit - 1) i0 = limit - 1;
    i1 = i0 + 1 < limit ? i0 + 1 : limit - 1;
    l1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void hbilinear_fwd_kernel(const bf16_t* __restrict__ x, int n, int h, int w, int c, int scale, bf16_t* __restrict__ y) {
    const int G = c / 8, H2 = h * scale, W2 = w * scale;
    const int64_t total = (int64_t)n * H2 * W2 * G;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % G) * 8;
    const int64_t pix = idx / G;
    const int ox = (int)(pix % W2), oy = (int)((pix / W2) % H2);
    const int64_t b = pix / ((int64_t)W2 * H2);
    int y0, y1, x0, x1; float ly, lx;
    hbilin_src(oy, scale, h, y0, y1, ly);
    hbilin_src(ox, scale, w, x0, x1, lx);
    const bf16_t* base = x + b * h * w * c + cc;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    unpack8(ld8(base + ((int64_t)y0 * w + x0) * c), v00); unpack8(ld8(base + ((int64_t)y0 * w + x1) * c), v01);
    unpack8(ld8(base + ((int64_t)y1 * w + x0) * c), v10); unpack8(ld8(base + ((int64_t)y1 * w + x1) * c), v11);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (1.f - ly) * ((1.f - lx) * v00[e] + lx * v01[e]) + ly * ((1.f - lx) * v10[e] + lx * v11[e]);
    st8_nt(y + pix * c + cc, pack8(o));
}
// gather form of the adjoint: an input pixel collects from the output rows / columns whose taps touch it
__global__ __launch_bounds__(256) void hbilinear_bwd_kernel(const bf16_t* __restrict__ dy, int n, int h, int w, int c, int scale, bf16_t* __restrict__ dx) {
    const int G = c / 8, H2 = h * scale, W2 = w * scale;
    const int64_t total = (int64_t)n * h * w * G;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % G) * 8;
    const int64_t pix = idx / G;
    const int ix = (int)(pix % w), iy = (int)((pix / w) % h);
    const int64_t b = pix / ((int64_t)w * h);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int oy_lo = (iy - 1) * scale < 0 ? 0 : (iy - 1) * scale;
    const int oy_hi = (iy + 2) * scale > H2 ? H2 : (iy + 2) * scale;
    const int ox_lo = (ix - 1) * scale < 0 ? 0 : (ix - 1) * scale;
    const int ox_hi = (ix + 2) * scale > W2 ? W2 : (ix + 2) * scale;
    for (int oy = oy_lo; oy < oy_hi; ++oy) {
        int y0, y1; float ly;
        hbilin_src(oy, scale, h, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox < ox_hi; ++ox) {
            int x0, x1; float lx;
            hbilin_src(ox, scale, w, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            if (wx == 0.f) continue;
            float g[8];
            unpack8(ld8(dy + ((b * H2 + oy) * W2 + ox) * c + cc), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wy * wx, g[e], acc[e]);
        }
    }
    st8_nt(dx + pix * c + cc, pack8(acc));
}

// ---- stem: fp32 image [n,h,w,c] (c <= 4) zero-padded by `pad` -> space-to-depth bf16 [n,(h+2pad)/2,(w+2pad)/2,16], channel
// order (row phase, column phase, c padded to 4): a stride-2 odd-k conv becomes a stride-1 valid conv (stem.hip, K4b) -------------
__global__ __launch_bounds__(256) void hstem_s2d_kernel(const float* __restrict__ x, int n, int h, int w, int c, int pad, int h2, int w2,
                                                        bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)n * h2 * w2 * 2;          // one row phase (8 channels) per thread
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int py = (int)(idx & 1);
    const int64_t pix = idx >> 1;
    const int ox = (int)(pix % w2), oy = (int)((pix / w2) % h2);
    const int64_t b = pix / ((int64_t)w2 * h2);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const int iy = 2 * oy + py - pad;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        const int ixx = 2 * ox + px - pad;
        if (iy >= 0 && iy < h && ixx >= 0 && ixx < w) {
            const float* p = x + ((b * h + iy) * (int64_t)w + ixx) * c;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < c) v[px * 4 + k] = p[k];
        }
    }
    st8_nt(out + pix * 16 + py * 8, pack8(v));
}

// ---- casts -----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hcast_from_f32_kernel(const float* __restrict__ src, int64_t n8, bf16_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const VecF<4> a = vload_nt<4>(src + i * 8), b = vload_nt<4>(src + i * 8 + 4);
    const float v[8] = {a.v[0], a.v[1], a.v[2], a.v[3], b.v[0], b.v[1], b.v[2], b.v[3]};
    st8_nt(dst + i * 8, pack8(v));
}
__global__ __launch_bounds__(256) void hcast_to_f32_kernel(const bf16_t* __restrict__ src, int64_t n8, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    unpack8(ld8_nt(src + i * 8), v);
    VecF<4> a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a.v[e] = v[e]; b.v[e] = v[4 + e]; }
    vstore_nt<4>(dst + i * 8, a);
    vstore_nt<4>(dst + i * 8 + 4, b);
}
// channel `ch` of a bf16 [m, c] matrix <-> an fp32 vector [m] (the 1-channel logits of a head padded to 8 channels)
__global__ __launch_bounds__(256) void hchannel_to_f32_kernel(const bf16_t* __restrict__ src, int64_t m, int c, int ch, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = bf16_lo((unsigned)src[i * c + ch]);
}
__global__ __launch_bounds__(256) void hchannel_from_f32_kernel(const float* __restrict__ src, int64_t m, int c, int ch, bf16_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one row octet per thread; every other channel is zero
    const int G = c / 8;
    if (i >= m * G) return;
    const int64_t row = i / G;
    const int c0 = (int)(i % G) * 8;
    const float sv = src[row];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e == ch) ? sv : 0.f;
    st8(dst + row * c + c0, pack8(v));
}

// partial rows of the strided reduction passes: ~256 K tasks (row, channel octet) whatever the channel count, so that narrow
// tensors ([2M, 64]: 8 octets per row) still fill the chip -- with tsii_common.h's partial_rows() that shape ran 128 blocks of
// 512 dependent iterations each (2.0 TB/s)
static inline int hbn_rows(int64_t m, int c) {
    int64_t r = 262144 / (c / 8);
    if (r > 32768) r = 32768;
    if (r > m) r = m;
    if (r < 1) r = 1;
    return (int)r;
}
static inline size_t hbn_coef_floats(int c) { return (size_t)6 * c + 8; }

}  // namespace tsii

using namespace tsii;

#define TSII_BF16_ELEM_CHECK(who, cond) TSII_REQUIRE(cond, who ": bad arguments (bf16 tensors: channel counts / element counts multiples of 8, 16-byte aligned)")

extern "C" int64_t tsii_bf16_bn_stat_rows(int64_t m, int c) { return (m > 0 && c > 0 && c % 8 == 0) ? hbn_rows(m, c) : 0; }

extern "C" int tsii_bf16_bn_stats(const uint16_t* y, int64_t m, int c, float* stat_part, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_bn_stats", y && stat_part && m > 0 && c > 0 && c % 8 == 0 && aligned16(y));
    const int R = hbn_rows(m, c);
    hipLaunchKernelGGL(hbn_stats_kernel, dim3(stream_grid((int64_t)R * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream, y, m, c, R, stat_part);
    return check_launch("bf16_bn_stats");
}

extern "C" int tsii_bf16_bn_act_fwd(const uint16_t* y, int64_t m, int c, const float* scale, const float* shift, int act, float slope,
                                    const uint16_t* residual, uint16_t* out, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_bn_act_fwd", y && scale && shift && out && m > 0 && c > 0 && c % 8 == 0 && aligned16(y) && aligned16(out) &&
                         aligned16(scale) && aligned16(shift) && (residual == nullptr || aligned16(residual)));
    TSII_REQUIRE(act >= 0 && act <= 4, "bf16_bn_act_fwd: unknown activation %d", act);
    hipLaunchKernelGGL(hbn_apply_kernel, dim3(flat_grid(m * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream, y, m, c, scale, shift, act, slope, residual, out);
    return check_launch("bf16_bn_act_fwd");
}

// workspace: [partial rows of the kernel's own reduction pass | level-1 sums of the row reduction] [coef table]
extern "C" size_t tsii_bf16_bn_ws_bytes(int64_t m, int c) {
    if (m <= 0 || c <= 0 || c % 8 != 0) return 0;
    const int R = hbn_rows(m, c);
    return (size_t)R * 2 * c * sizeof(float) + bn_bwd_reduce_ws_bytes(R, c) + hbn_coef_floats(c) * sizeof(float) + 64;
}

extern "C" int tsii_bf16_bn_act_bwd(const uint16_t* dout, const uint16_t* y, int64_t m, int c, const float* mean, const float* var,
                                    const float* gamma, const float* beta, float eps, int act, float slope, int training,
                                    const float* bwd_part, int64_t rows, uint16_t* dy, float* dgamma, float* dbeta,
                                    void* ws, size_t ws_bytes, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_bn_act_bwd", dout && y && mean && var && gamma && beta && dy && dgamma && dbeta && ws && m > 0 && c > 0 && c % 8 == 0 &&
                         aligned16(dout) && aligned16(y) && aligned16(dy));
    TSII_REQUIRE(ws_bytes >= tsii_bf16_bn_ws_bytes(m, c), "bf16_bn_act_bwd: workspace too small (tsii_bf16_bn_ws_bytes)");
    TSII_REQUIRE(bwd_part == nullptr || (rows > 0 && rows < (1ll << 31)), "bf16_bn_act_bwd: bad partial-row count");
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
    const size_t usable = ws_bytes - (size_t)(base - reinterpret_cast<char*>(ws));
    const size_t coef_bytes = ((hbn_coef_floats(c) * sizeof(float)) + 15) & ~(size_t)15;
    float* coef = reinterpret_cast<float*>(base + ((usable - coef_bytes) & ~(size_t)15));      // table at the end, 16-byte aligned
    int rc;
    if (bwd_part == nullptr) {
        const int R = hbn_rows(m, c);
        float* part = reinterpret_cast<float*>(base);                          // [R][2][c]
        void* l1 = base + (size_t)R * 2 * c * sizeof(float);
        hipLaunchKernelGGL(hbn_bwd_partial_kernel, dim3(stream_grid((int64_t)R * (c / 8), 256)), dim3(256), 0, st, dout, y, m, c, mean, var, gamma, beta,
                           eps, act, slope, R, part);
        rc = check_launch("bf16_bn_bwd_partial");
        if (rc) return rc;
        rc = launch_bn_bwd_reduce(mean, var, gamma, beta, eps, training, part, R, m, c, dgamma, dbeta, l1, coef, st);
    } else {
        TSII_REQUIRE(bn_bwd_reduce_ws_bytes(rows, c) + coef_bytes + 32 <= usable, "bf16_bn_act_bwd: workspace too small for %lld partial rows", (long long)rows);
        rc = launch_bn_bwd_reduce(mean, var, gamma, beta, eps, training, bwd_part, rows, m, c, dgamma, dbeta, base, coef, st);
    }
    if (rc) return rc;
#ifndef HBN_APPLY_RPT
#define HBN_APPLY_RPT 4          // rows per thread on the large tensors (measured: 8 -> 79.8, 4 -> 73.1, 1 -> 110.3 us on [131072, 512] incl. the reduction)
#endif
    const int64_t vecs = m * (c / 8);
    if (vecs >= (1ll << 21)) hipLaunchKernelGGL(hbn_bwd_apply_kernel<HBN_APPLY_RPT>, dim3(flat_grid(cdiv64(m, HBN_APPLY_RPT) * (c / 8), 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    else if (vecs >= (1ll << 19)) hipLaunchKernelGGL(hbn_bwd_apply_kernel<2>, dim3(flat_grid(cdiv64(m, 2) * (c / 8), 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    else hipLaunchKernelGGL(hbn_bwd_apply_kernel<1>, dim3(flat_grid(vecs, 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    return check_launch("bf16_bn_bwd_apply");
}

extern "C" int tsii_bf16_add_act_fwd(const uint16_t* a, const uint16_t* b, int64_t numel, int act, float slope, uint16_t* out, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_add_act_fwd", a && b && out && numel > 0 && numel % 8 == 0 && aligned16(a) && aligned16(b) && aligned16(out));
    TSII_REQUIRE(act >= 0 && act <= 4, "bf16_add_act_fwd: unknown activation %d", act);
    hipLaunchKernelGGL(hadd_act_kernel, dim3(flat_grid(numel / 8, 256)), dim3(256), 0, (hipStream_t)stream, a, b, numel / 8, act, slope, out);
    return check_launch("bf16_add_act_fwd");
}

extern "C" int tsii_bf16_act_bwd(const uint16_t* dout, const uint16_t* x, int64_t numel, int act, float slope, uint16_t* dx, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_act_bwd", dout && x && dx && numel > 0 && numel % 8 == 0 && aligned16(dout) && aligned16(x) && aligned16(dx));
    hipLaunchKernelGGL(hact_bwd_kernel, dim3(flat_grid(numel / 8, 256)), dim3(256), 0, (hipStream_t)stream, dout, x, numel / 8, act, slope, dx);
    return check_launch("bf16_act_bwd");
}

extern "C" int tsii_bf16_copy_channels(uint16_t* big, int64_t m, int cbig, int coff, uint16_t* small_, int csmall, int to_dst, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_copy_channels", big && small_ && m > 0 && cbig > 0 && csmall > 0 && coff >= 0 && coff + csmall <= cbig &&
                         cbig % 8 == 0 && csmall % 8 == 0 && coff % 8 == 0 && aligned16(big) && aligned16(small_));
    hipLaunchKernelGGL(hcopy_channels_kernel, dim3(flat_grid(m * (csmall / 8), 256)), dim3(256), 0, (hipStream_t)stream, big, m, cbig, coff, small_, csmall, to_dst);
    return check_launch("bf16_copy_channels");
}

extern "C" int tsii_bf16_bilinear_up_fwd(const uint16_t* x, int n, int h, int w, int c, int scale, uint16_t* y, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_bilinear_up_fwd", x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && scale >= 1 && aligned16(x) && aligned16(y));
    const int64_t total = (int64_t)n * h * scale * w * scale * (c / 8);
    hipLaunchKernelGGL(hbilinear_fwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, scale, y);
    return check_launch("bf16_bilinear_up_fwd");
}

extern "C" int tsii_bf16_bilinear_up_bwd(const uint16_t* dy, int n, int h, int w, int c, int scale, uint16_t* dx, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_bilinear_up_bwd", dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && scale >= 1 && aligned16(dx) && aligned16(dy));
    const int64_t total = (int64_t)n * h * w * (c / 8);
    hipLaunchKernelGGL(hbilinear_bwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, n, h, w, c, scale, dx);
    return check_launch("bf16_bilinear_up_bwd");
}

extern "C" int tsii_bf16_stem_s2d(const float* x, int n, int h, int w, int c, int pad, uint16_t* out, void* stream) {
    TSII_REQUIRE(x && out && n > 0 && h > 0 && w > 0 && c > 0 && c <= 4 && pad >= 0 && (h + 2 * pad) % 2 == 0 && (w + 2 * pad) % 2 == 0 && aligned16(out),
                 "bf16_stem_s2d: bad arguments (c <= 4, even padded image)");
    const int h2 = (h + 2 * pad) / 2, w2 = (w + 2 * pad) / 2;
    hipLaunchKernelGGL(hstem_s2d_kernel, dim3(flat_grid((int64_t)n * h2 * w2 * 2, 256)), dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, pad, h2, w2, out);
    return check_launch("bf16_stem_s2d");
}

extern "C" int tsii_bf16_from_f32(const float* src, int64_t numel, uint16_t* dst, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_from_f32", src && dst && numel > 0 && numel % 8 == 0 && aligned16(src) && aligned16(dst));
    hipLaunchKernelGGL(hcast_from_f32_kernel, dim3(flat_grid(numel / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, numel / 8, dst);
    return check_launch("bf16_from_f32");
}

extern "C" int tsii_bf16_to_f32(const uint16_t* src, int64_t numel, float* dst, void* stream) {
    TSII_BF16_ELEM_CHECK("bf16_to_f32", src && dst && numel > 0 && numel % 8 == 0 && aligned16(src) && aligned16(dst));
    hipLaunchKernelGGL(hcast_to_f32_kernel, dim3(flat_grid(numel / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, numel / 8, dst);
    return check_launch("bf16_to_f32");
}

extern "C" int tsii_bf16_channel_to_f32(const uint16_t* src, int64_t m, int c, int ch, float* dst, void* stream) {
    TSII_REQUIRE(src && dst && m > 0 && c > 0 && ch >= 0 && ch < c, "bf16_channel_to_f32: bad arguments");
    hipLaunchKernelGGL(hchannel_to_f32_kernel, dim3(flat_grid(m, 256)), dim3(256), 0, (hipStream_t)stream, src, m, c, ch, dst);
    return check_launch("bf16_channel_to_f32");
}

extern "C" int tsii_bf16_channel_from_f32(const float* src, int64_t m, int c, int ch, uint16_t* dst, void* stream) {
    TSII_REQUIRE(src && dst && m > 0 && c > 0 && c % 8 == 0 && ch >= 0 && ch < c && aligned16(dst), "bf16_channel_from_f32: bad arguments");
    hipLaunchKernelGGL(hchannel_from_f32_kernel, dim3(flat_grid(m * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream, src, m, c, ch, dst);
    return check_launch("bf16_channel_from_f32");
}
