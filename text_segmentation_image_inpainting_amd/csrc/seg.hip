// Kernels of the text-segmentation path that are not convolutions / BatchNorm: residual add +
// activation, channel concat, bilinear up-sampling, global average pool and the scSE combine
// (models/common.py:13-43, models/text_segmentation.py:60-84,104-114), and BinaryFocalLoss (loss.py:58-75).
// All HBM-bound streaming / small reductions on NHWC rows.
#include "tsii_common.h"

namespace tsii {

template <int W>
__global__ void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n4, int act,
                               float slope, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        VecF<W> u = vload<W>(a + i * W);
        const VecF<W> v = vload<W>(b + i * W);
#pragma unroll
        for (int e = 0; e < W; ++e) u.v[e] = apply_act(u.v[e] + v.v[e], act, slope);
        vstore<W>(out + i * W, u);
    }
}

// launched with flat_grid() (one vector per thread; any grid is correct)
template <int W>
__global__ void copy_channels_kernel(float* __restrict__ big, int64_t m, int cbig, int coff, float* __restrict__ sm,
                                     int csm, int to_dst) {
    const unsigned CG = (unsigned)(csm / W);
    const int64_t total = m * CG;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int c = (int)(gt % CG) * W;
    const int64_t rstep = stride / CG;
    int64_t row = gt / CG;
    for (int64_t idx = gt; idx < total; idx += stride, row += rstep) {
        float* pb = big + row * cbig + coff + c;
        float* ps = sm + row * csm + c;
        if (to_dst) vstore<W>(pb, vload<W>(ps));
        else vstore<W>(ps, vload<W>(pb));
    }
}

__device__ __forceinline__ void bilin_src(int o, int scale, int limit, int& i0, int& i1, float& l1) {
    float s = ((float)o + 0.5f) / (float)scale - 0.5f;   // align_corners = False
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > limit - 1) i0 = limit - 1;
    i1 = i0 + 1 < limit ? i0 + 1 : limit - 1;
    l1 = s - (float)i0;
}

template <int W>
__global__ void bilinear_up_fwd_kernel(const float* __restrict__ x, int n, int h, int w, int c, int scale,
                                       float* __restrict__ y) {
    const int CG = c / W, H2 = h * scale, W2 = w * scale;
    const int64_t total = (int64_t)n * H2 * W2 * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int ox = (int)(pix % W2), oy = (int)((pix / W2) % H2);
        const int64_t b = pix / ((int64_t)W2 * H2);
        int y0, y1, x0, x1; float ly, lx;
        bilin_src(oy, scale, h, y0, y1, ly);
        bilin_src(ox, scale, w, x0, x1, lx);
        const float* base = x + b * h * w * c + cc;
        const VecF<W> v00 = vload<W>(base + ((int64_t)y0 * w + x0) * c), v01 = vload<W>(base + ((int64_t)y0 * w + x1) * c);
        const VecF<W> v10 = vload<W>(base + ((int64_t)y1 * w + x0) * c), v11 = vload<W>(base + ((int64_t)y1 * w + x1) * c);
        VecF<W> o;
#pragma unroll
        for (int e = 0; e < W; ++e)
            o.v[e] = (1.f - ly) * ((1.f - lx) * v00.v[e] + lx * v01.v[e]) + ly * ((1.f - lx) * v10.v[e] + lx * v11.v[e]);
        vstore<W>(y + pix * c + cc, o);
    }
}

// gather form of the adjoint: an input pixel collects from the output rows/cols whose taps touch it
template <int W>
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dy, int n, int h, int w, int c, int scale,
                                       float* __restrict__ dx) {
    const int CG = c / W, H2 = h * scale, W2 = w * scale;
    const int64_t total = (int64_t)n * h * w * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int ix = (int)(pix % w), iy = (int)((pix / w) % h);
        const int64_t b = pix / ((int64_t)w * h);
        VecF<W> acc;
#pragma unroll
        for (int e = 0; e < W; ++e) acc.v[e] = 0.f;
        const int oy_lo = (iy - 1) * scale < 0 ? 0 : (iy - 1) * scale;
        const int oy_hi = (iy + 2) * scale > H2 ? H2 : (iy + 2) * scale;
        const int ox_lo = (ix - 1) * scale < 0 ? 0 : (ix - 1) * scale;
        const int ox_hi = (ix + 2) * scale > W2 ? W2 : (ix + 2) * scale;
        for (int oy = oy_lo; oy < oy_hi; ++oy) {
            int y0, y1; float ly;
            bilin_src(oy, scale, h, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox < ox_hi; ++ox) {
                int x0, x1; float lx;
                bilin_src(ox, scale, w, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                const VecF<W> g = vload<W>(dy + ((b * H2 + oy) * W2 + ox) * c + cc);
#pragma unroll
                for (int e = 0; e < W; ++e) acc.v[e] = fmaf(wy * wx, g.v[e], acc.v[e]);
            }
        }
        vstore<W>(dx + pix * c + cc, acc);
    }
}

// per-sample column sums of a (optionally a*b): part[n][chunk][c]; block = NB columns x L row lanes
__global__ __launch_bounds__(256) void sample_colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int hw, int c, int NB, int L, int rows_per_block, int chunks,
                                                            float* __restrict__ part) {
    __shared__ float sh[256];
    const int n = blockIdx.z, chunk = blockIdx.x;
    const int cl = threadIdx.x % NB, lane = threadIdx.x / NB;
    const int col = blockIdx.y * NB + cl;
    const int r0 = chunk * rows_per_block;
    const int r1 = r0 + rows_per_block < hw ? r0 + rows_per_block : hw;
    float s = 0.f;
    if (lane < L && col < c) {
        const int64_t base = (int64_t)n * hw * c;
        for (int r = r0 + lane; r < r1; r += L) {
            float v = a[base + (int64_t)r * c + col];
            if (b != nullptr) v *= b[base + (int64_t)r * c + col];
            s += v;
        }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && col < c) {
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += sh[l * NB + cl];
        part[((int64_t)n * chunks + chunk) * c + col] = t;
    }
}
__global__ void sample_colsum_final_kernel(const float* __restrict__ part, int n, int chunks, int c, float scale,
                                           float* __restrict__ out) {
    const int total = n * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / c, col = i % c;
        double s = 0.0;
        for (int k = 0; k < chunks; ++k) s += (double)part[((int64_t)b * chunks + k) * c + col];
        out[i] = (float)(s * (double)scale);
    }
}
// the same partial sums with 16-byte loads (c % 4 == 0, 16-byte aligned operands; round 6): a thread owns one channel quad of NB4
// <= 256 quads and every L-th row of the chunk, two rows in flight -- the 4-byte form above ran at 3 TB/s on the pooling of scSE
template <bool HASB>
__global__ __launch_bounds__(256) void sample_colsum4_kernel(const float* __restrict__ a, const float* __restrict__ b, int hw, int c, int NB4, int L,
                                                             int rows_per_block, int chunks, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float sh[256 * 4];
    const int n = blockIdx.z, chunk = blockIdx.x;
    const int cl = threadIdx.x % NB4, lane = threadIdx.x / NB4;
    const int cq = blockIdx.y * NB4 + cl;
    const int r0 = chunk * rows_per_block;
    const int r1 = r0 + rows_per_block < hw ? r0 + rows_per_block : hw;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    const bool act = lane < L && cq * 4 < c;
    if (act) {
        const float* pa = a + (int64_t)n * hw * c + cq * 4;
        const float* pb = HASB ? b + (int64_t)n * hw * c + cq * 4 : nullptr;
        int r = r0 + lane;
        for (; r + L < r1; r += 2 * L) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(pa + (int64_t)r * c), v1 = *reinterpret_cast<const f32x4*>(pa + (int64_t)(r + L) * c);
            if (HASB) { v0 *= *reinterpret_cast<const f32x4*>(pb + (int64_t)r * c); v1 *= *reinterpret_cast<const f32x4*>(pb + (int64_t)(r + L) * c); }
            s0 += v0; s1 += v1;
        }
        if (r < r1) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(pa + (int64_t)r * c);
            if (HASB) v0 *= *reinterpret_cast<const f32x4*>(pb + (int64_t)r * c);
            s0 += v0;
        }
    }
    *reinterpret_cast<f32x4*>(sh + threadIdx.x * 4) = s0 + s1;
    __syncthreads();
    if (lane == 0 && cq * 4 < c) {
        f32x4 tsum = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) tsum += *reinterpret_cast<const f32x4*>(sh + (l * NB4 + cl) * 4);
        *reinterpret_cast<f32x4*>(part + ((int64_t)n * chunks + chunk) * c + cq * 4) = tsum;
    }
}
static inline int sc_chunks(int hw) { int k = (hw + 255) / 256; return k > 64 ? 64 : (k < 1 ? 1 : k); }
static int launch_sample_colsum(const float* a, const float* b, int n, int hw, int c, float scale, float* out, float* ws,
                                hipStream_t st) {
    const int chunks = sc_chunks(hw);
    const int rpb = cdiv(hw, chunks);
    if (c % 4 == 0 && aligned16(a) && (b == nullptr || aligned16(b)) && aligned16(ws)) {
        const int CG = c / 4, NB4 = CG < 256 ? CG : 256, L4 = 256 / NB4;
        const dim3 grid4(chunks, cdiv(CG, NB4), n);
        if (b != nullptr) hipLaunchKernelGGL((sample_colsum4_kernel<true>), grid4, dim3(256), 0, st, a, b, hw, c, NB4, L4, rpb, chunks, ws);
        else hipLaunchKernelGGL((sample_colsum4_kernel<false>), grid4, dim3(256), 0, st, a, b, hw, c, NB4, L4, rpb, chunks, ws);
        int rc4 = check_launch("sample_colsum4");
        if (rc4) return rc4;
        hipLaunchKernelGGL(sample_colsum_final_kernel, dim3(cdiv(n * c, 256)), dim3(256), 0, st, ws, n, chunks, c, scale, out);
        return check_launch("sample_colsum_final");
    }
    const int NB = c < 256 ? c : 256, L = 256 / NB;
    hipLaunchKernelGGL(sample_colsum_kernel, dim3(chunks, cdiv(c, NB), n), dim3(256), 0, st, a, b, hw, c, NB, L, rpb, chunks, ws);
    int rc = check_launch("sample_colsum");
    if (rc) return rc;
    hipLaunchKernelGGL(sample_colsum_final_kernel, dim3(cdiv(n * c, 256)), dim3(256), 0, st, ws, n, chunks, c, scale, out);
    return check_launch("sample_colsum_final");
}

template <int W>
__global__ void gap_bwd_kernel(const float* __restrict__ dgap, int n, int hw, int c, float* __restrict__ dx) {
    const int CG = c / W;
    const int64_t total = (int64_t)n * hw * CG;
    const float inv = 1.f / (float)hw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int64_t b = pix / hw;
        VecF<W> g = vload<W>(dgap + b * c + cc);
#pragma unroll
        for (int e = 0; e < W; ++e) g.v[e] *= inv;
        vstore<W>(dx + pix * c + cc, g);
    }
}

// out = x * (cse[n,c] + sse[n,hw])   (g == nullptr: forward)  |  dx = g * (cse + sse)
template <int W>
__global__ void scse_scale_kernel(const float* __restrict__ x, const float* __restrict__ cse, const float* __restrict__ sse,
                                  int n, int hw, int c, float* __restrict__ out) {
    const int CG = c / W;
    const int64_t total = (int64_t)n * hw * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int64_t b = pix / hw;
        VecF<W> v = vload<W>(x + pix * c + cc);
        const VecF<W> cs = vload<W>(cse + b * c + cc);
        const float ss = sse[pix];
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] = v.v[e] * cs.v[e] + v.v[e] * ss;   // mul, mul, add like the reference
        vstore<W>(out + pix * c + cc, v);
    }
}
// dsse[pix] = sum_c g*x : one wave per pixel, lanes stride the channels, shuffle reduce
__global__ __launch_bounds__(256) void scse_dsse_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        int64_t npix, int c, float* __restrict__ dsse) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pix = wave; pix < npix; pix += nwaves) {
        float s = 0.f;
        for (int k = lane; k < c; k += 64) s = fmaf(g[pix * c + k], x[pix * c + k], s);
        s = wave_sum(s);
        if (lane == 0) dsse[pix] = s;
    }
}

// scSE backward in ONE pass (round 6): dx = g * (cse + sse), dsse[pix] = sum_c g x, dcse[n, c] = sum_pix g x.  The three-kernel form
// above reads g three times and x twice (dsse with 4-byte loads, one wave per pixel): 2.5 TB/s of algorithmic traffic, 7.8 ms of
// TextSegament's 512^2 bs-64 step.  Here a group of G lanes (G = the channel quads of a pixel rounded up to a power of two, at most
// 64) owns a pixel at a time: 16-byte loads of g and x, the pixel's sum by shuffles inside the group, the channel sums in registers
// (Q quads per lane) over the block's pixel range, combined through LDS into one partial row per (image, chunk) -- the layout
// sample_colsum_final_kernel reduces.
template <int Q>
__global__ __launch_bounds__(256) void scse_bwd_fused_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ cse,
                                                             const float* __restrict__ sse, int hw, int c, int G, int rows_per_block, int chunks,
                                                             float* __restrict__ dx, float* __restrict__ dsse, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float sh[8192];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane % G, gp = lane / G, PPW = 64 / G, SLOTS = 4 * PPW;
    const int CG = c >> 2;
    const int r0 = chunk * rows_per_block;
    const int r1 = r0 + rows_per_block < hw ? r0 + rows_per_block : hw;
    const int64_t base = (int64_t)n * hw;
    f32x4 cs[Q], acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int cq = gl + q * G;
        cs[q] = cq < CG ? *reinterpret_cast<const f32x4*>(cse + (int64_t)n * c + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int iters = (r1 - r0 + SLOTS - 1) / SLOTS;             // block-uniform: every lane takes part in every shuffle
    for (int it = 0; it < iters; ++it) {
        const int p = r0 + it * SLOTS + wave * PPW + gp;
        const bool ok = p < r1;
        const int64_t pix = base + (ok ? p : r1 - 1);
        const float ss = sse[pix];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int cq = gl + q * G;
            if (cq < CG) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + pix * c + cq * 4), xv = *reinterpret_cast<const f32x4*>(x + pix * c + cq * 4);
                f32x4 pr = gv * xv;
                if (!ok) pr = f32x4{0.f, 0.f, 0.f, 0.f};
                s += (pr.x + pr.y) + (pr.z + pr.w);
                acc[q] += pr;
                if (ok) {
                    const f32x4 o = gv * cs[q] + gv * ss;            // mul, mul, add like the reference (and scse_scale_kernel)
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dx + pix * c + cq * 4));
                }
            }
        }
        for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m, 64);
        if (gl == 0 && ok) dsse[pix] = s;
    }
    // channel sums of the block: [slot][c] through LDS
    const int slot = wave * PPW + gp;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int cq = gl + q * G;
        if (cq < CG) *reinterpret_cast<f32x4*>(sh + slot * c + cq * 4) = acc[q];
    }
    __syncthreads();
    float* prow = part + ((int64_t)n * chunks + chunk) * c;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float sum = 0.f;
        for (int sl = 0; sl < SLOTS; ++sl) sum += sh[sl * c + ch];
        prow[ch] = sum;
    }
}

// pixel shuffle (K12): pure permutation; lanes walk the OUTPUT channels so the wide side is coalesced
__global__ void pixel_shuffle_kernel(const float* __restrict__ src, int64_t total, int h, int w, int c, int r, int inverse,
                                     float* __restrict__ dst) {
    const int hr = h * r, wr = w * r, crr = c * r * r;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int64_t opix = idx / c;
        const int ox = (int)(opix % wr), oy = (int)((opix / wr) % hr);
        const int64_t b = opix / ((int64_t)wr * hr);
        const int64_t lo = ((b * h + oy / r) * w + ox / r) * crr + (int64_t)ch * r * r + (oy % r) * r + (ox % r);
        if (inverse) dst[lo] = src[idx];
        else dst[idx] = src[lo];
    }
}

// BinaryFocalLoss element (loss.py:66-75)
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float focal_elem(float x, float t, float gamma, float bw, float ww) {
    const float w = t > 0.f ? ww : bw;
    const float u = -x * (t * 2.f - 1.f);
    const float pt = (u < 0.f ? u : 0.f) - softplus_neg_abs(u);               // logsigmoid(u)
    const float bce = w * ((x > 0.f ? x : 0.f) - x * t + softplus_neg_abs(x));
    return expf(pt * gamma) * bce;
}
__global__ __launch_bounds__(256) void bce_focal_partial_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                                int64_t numel, float gamma, float bw, float ww,
                                                                float* __restrict__ part) {
    __shared__ float wsum[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        s += focal_elem(x[i], t[i], gamma, bw, ww);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ void mean_final_kernel(const float* __restrict__ part, int nblocks, int64_t numel, float* __restrict__ loss) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblocks; ++i) s += (double)part[i];
        loss[0] = (float)(s / (double)numel);
    }
}
__global__ void bce_focal_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t numel, float gamma,
                                     float bw, float ww, const float* __restrict__ gscale, float* __restrict__ dx) {
    const float gs = gscale[0] / (float)numel;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = x[i], tv = t[i];
        const float w = tv > 0.f ? ww : bw;
        const float sgn = tv * 2.f - 1.f;
        const float u = -xv * sgn;
        const float pt = (u < 0.f ? u : 0.f) - softplus_neg_abs(u);
        const float sig_u = 1.f / (1.f + expf(-u));
        const float dpt = -sgn * (1.f - sig_u);
        const float bce = w * ((xv > 0.f ? xv : 0.f) - xv * tv + softplus_neg_abs(xv));
        const float dbce = w * (1.f / (1.f + expf(-xv)) - tv);
        dx[i] = gs * expf(pt * gamma) * (gamma * dpt * bce + dbce);
    }
}

}  // namespace tsii

using namespace tsii;

extern "C" int tsii_pixel_shuffle(const float* src, int n, int h, int w, int c, int r, int inverse, float* dst, void* stream) {
    TSII_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && r >= 1, "pixel_shuffle: bad arguments");
    const int64_t total = (int64_t)n * h * r * w * r * c;
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, total, h, w, c, r,
                       inverse, dst);
    return check_launch("pixel_shuffle");
}

extern "C" int tsii_add_act_fwd(const float* a, const float* b, int64_t numel, int act, float slope, float* out, void* stream) {
    TSII_REQUIRE(a && b && out && numel > 0, "add_act_fwd: bad arguments");
    TSII_REQUIRE(act >= 0 && act <= 4, "add_act_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    if (numel % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(out))
        hipLaunchKernelGGL((add_act_kernel<4>), dim3(flat_grid(numel / 4, 256)), dim3(256), 0, st, a, b, numel / 4, act, slope, out);
    else
        hipLaunchKernelGGL((add_act_kernel<1>), dim3(flat_grid(numel, 256)), dim3(256), 0, st, a, b, numel, act, slope, out);
    return check_launch("add_act_fwd");
}

extern "C" int tsii_copy_channels(float* big, int64_t m, int cbig, int coff, float* small_, int csmall, int to_dst, void* stream) {
    TSII_REQUIRE(big && small_ && m > 0 && cbig > 0 && csmall > 0 && coff >= 0 && coff + csmall <= cbig, "copy_channels: bad arguments");
    const bool vec = (cbig % 4 == 0) && (csmall % 4 == 0) && (coff % 4 == 0) && aligned16(big) && aligned16(small_);
    const int CG = vec ? csmall / 4 : csmall;
    const unsigned grid = flat_grid(m * CG, 256);
    if (vec) hipLaunchKernelGGL((copy_channels_kernel<4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, big, m, cbig, coff, small_, csmall, to_dst);
    else hipLaunchKernelGGL((copy_channels_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, big, m, cbig, coff, small_, csmall, to_dst);
    return check_launch("copy_channels");
}

extern "C" int tsii_bilinear_up_fwd(const float* x, int n, int h, int w, int c, int scale, float* y, void* stream) {
    TSII_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && scale >= 1, "bilinear_up_fwd: bad arguments");
    const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(y);
    const int64_t total = (int64_t)n * h * scale * w * scale * (vec ? c / 4 : c);
    if (vec) hipLaunchKernelGGL((bilinear_up_fwd_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, scale, y);
    else hipLaunchKernelGGL((bilinear_up_fwd_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, n, h, w, c, scale, y);
    return check_launch("bilinear_up_fwd");
}

extern "C" int tsii_bilinear_up_bwd(const float* dy, int n, int h, int w, int c, int scale, float* dx, void* stream) {
    TSII_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && scale >= 1, "bilinear_up_bwd: bad arguments");
    const bool vec = (c % 4 == 0) && aligned16(dx) && aligned16(dy);
    const int64_t total = (int64_t)n * h * w * (vec ? c / 4 : c);
    if (vec) hipLaunchKernelGGL((bilinear_up_bwd_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, n, h, w, c, scale, dx);
    else hipLaunchKernelGGL((bilinear_up_bwd_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, n, h, w, c, scale, dx);
    return check_launch("bilinear_up_bwd");
}

extern "C" size_t tsii_gap_ws_bytes(int n, int hw, int c) {
    if (n <= 0 || hw <= 0 || c <= 0) return 0;
    return (size_t)n * sc_chunks(hw) * c * sizeof(float);
}

extern "C" int tsii_gap_fwd(const float* x, int n, int hw, int c, float* gap, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && gap && ws && n > 0 && hw > 0 && c > 0, "gap_fwd: bad arguments");
    TSII_REQUIRE(ws_bytes >= tsii_gap_ws_bytes(n, hw, c) && n <= 65535, "gap_fwd: workspace too small / batch too large");
    return launch_sample_colsum(x, nullptr, n, hw, c, 1.f / (float)hw, gap, (float*)ws, (hipStream_t)stream);
}

extern "C" int tsii_gap_bwd(const float* dgap, int n, int hw, int c, float* dx, void* stream) {
    TSII_REQUIRE(dgap && dx && n > 0 && hw > 0 && c > 0, "gap_bwd: bad arguments");
    const bool vec = (c % 4 == 0) && aligned16(dgap) && aligned16(dx);
    const int64_t total = (int64_t)n * hw * (vec ? c / 4 : c);
    if (vec) hipLaunchKernelGGL((gap_bwd_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dgap, n, hw, c, dx);
    else hipLaunchKernelGGL((gap_bwd_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dgap, n, hw, c, dx);
    return check_launch("gap_bwd");
}

extern "C" int tsii_scse_fwd(const float* x, const float* cse, const float* sse, int n, int hw, int c, float* out, void* stream) {
    TSII_REQUIRE(x && cse && sse && out && n > 0 && hw > 0 && c > 0, "scse_fwd: bad arguments");
    const bool vec = (c % 4 == 0) && aligned16(x) && aligned16(cse) && aligned16(out);
    const int64_t total = (int64_t)n * hw * (vec ? c / 4 : c);
    if (vec) hipLaunchKernelGGL((scse_scale_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cse, sse, n, hw, c, out);
    else hipLaunchKernelGGL((scse_scale_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cse, sse, n, hw, c, out);
    return check_launch("scse_fwd");
}

extern "C" int tsii_scse_bwd(const float* g, const float* x, const float* cse, const float* sse, int n, int hw, int c,
                             float* dx, float* dcse, float* dsse, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(g && x && cse && sse && dx && dcse && dsse && ws && n > 0 && hw > 0 && c > 0, "scse_bwd: bad arguments");
    TSII_REQUIRE(ws_bytes >= tsii_gap_ws_bytes(n, hw, c) && n <= 65535, "scse_bwd: workspace too small / batch too large");
    hipStream_t st = (hipStream_t)stream;
    // one pass over g and x (scse_bwd_fused_kernel) when the channel quads of a pixel fit 64 lanes x 8 and the block's channel sums
    // fit its LDS buffer; else the three-kernel form
    {
        const int CG = c / 4;
        int G = 1;
        while (G < CG && G < 64) G <<= 1;
        const int Q = cdiv(CG, G), slots = 4 * (64 / G);
        if (c % 4 == 0 && Q <= 8 && (int64_t)slots * c <= 8192 && aligned16(g) && aligned16(x) && aligned16(cse) && aligned16(dx)) {
            const int chunks = sc_chunks(hw), rpb = cdiv(hw, chunks);
            const dim3 grid(chunks, n);
#define TSII_SCSE_BWD(QV) hipLaunchKernelGGL((scse_bwd_fused_kernel<QV>), grid, dim3(256), 0, st, g, x, cse, sse, hw, c, G, rpb, chunks, dx, dsse, (float*)ws)
            if (Q <= 1) TSII_SCSE_BWD(1); else if (Q <= 2) TSII_SCSE_BWD(2); else if (Q <= 4) TSII_SCSE_BWD(4); else TSII_SCSE_BWD(8);
#undef TSII_SCSE_BWD
            int rcf = check_launch("scse_bwd_fused");
            if (rcf) return rcf;
            hipLaunchKernelGGL(sample_colsum_final_kernel, dim3(cdiv(n * c, 256)), dim3(256), 0, st, (const float*)ws, n, chunks, c, 1.f, dcse);
            return check_launch("sample_colsum_final");
        }
    }
    int rc = tsii_scse_fwd(g, cse, sse, n, hw, c, dx, stream);   // dx = g*cse + g*sse
    if (rc) return rc;
    rc = launch_sample_colsum(g, x, n, hw, c, 1.f, dcse, (float*)ws, st);
    if (rc) return rc;
    const int64_t npix = (int64_t)n * hw;
    hipLaunchKernelGGL(scse_dsse_kernel, dim3(stream_grid(npix * 64, 256)), dim3(256), 0, st, g, x, npix, c, dsse);
    return check_launch("scse_dsse");
}

extern "C" int tsii_bce_focal_fwd(const float* x, const float* t, int64_t numel, float gamma, float background_w,
                                  float words_w, float* loss, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && t && loss && ws && numel > 0, "bce_focal_fwd: bad arguments");
    TSII_REQUIRE(ws_bytes >= tsii_l1_ws_bytes(numel), "bce_focal_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = cdiv64(numel, 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(bce_focal_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, t, numel, gamma, background_w, words_w, (float*)ws);
    int rc = check_launch("bce_focal_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, (int)nb, numel, loss);
    return check_launch("bce_focal_final");
}

extern "C" int tsii_bce_focal_bwd(const float* x, const float* t, int64_t numel, float gamma, float background_w,
                                  float words_w, const float* gscale, float* dx, void* stream) {
    TSII_REQUIRE(x && t && gscale && dx && numel > 0, "bce_focal_bwd: bad arguments");
    hipLaunchKernelGGL(bce_focal_bwd_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, x, t, numel,
                       gamma, background_w, words_w, gscale, dx);
    return check_launch("bce_focal_bwd");
}
