// K4 (general path): dense k x k partial convolution with groups == 1
// (PartialConv.forward, models/partial_convolution.py:49-80), any Cin/Cout/kernel/stride/dilation.
//
// Direct NHWC convolution on the vector ALUs.  In ImageFill it serves the two layers whose
// shapes do not suit the matrix cores: the 7x7 s2 stem on 3 channels with a per-channel mask
// (models/image_inpainting.py:23) and the final 3x3 35->3 layer (:44) -- together 3% of the
// model's MACs.  One thread produces 4 output channels of one pixel: the input value is a
// broadcast load, the weights (re-laid out once per call as [tap][ci][co]) are 16-byte loads.
// x*mask (:51) comes either from a full per-channel mask or from the two-plane row scale;
// count division / hole zeroing (:66-72) use the K1 planes.
#include "tsii_common.h"

namespace tsii {

struct ConvGeom {
    int n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo;
};

__device__ __forceinline__ float in_mask(const float* __restrict__ mfull, const RowScale& rs, int64_t ipix, int cin, int ci) {
    if (mfull != nullptr) return mfull[ipix * cin + ci];
    return row_scale_at(rs, ipix, ci);
}

// w[co][ci][t] -> wf[(t*cin + ci)*coutp + co], zero padded to coutp = 4*ceil(cout/4)
__global__ void dense_prep_fwd_kernel(const float* __restrict__ w, int cin, int cout, int T, int coutp,
                                      float* __restrict__ wf) {
    const int64_t total = (int64_t)T * cin * coutp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % coutp);
        const int64_t k = i / coutp;
        const int ci = (int)(k % cin);
        const int t = (int)(k / cin);
        wf[i] = co < cout ? w[((int64_t)co * cin + ci) * T + t] : 0.f;
    }
}
// w[co][ci][t] -> wb[(t*cout + co)*cinp + ci], zero padded
__global__ void dense_prep_dx_kernel(const float* __restrict__ w, int cin, int cout, int T, int cinp,
                                     float* __restrict__ wb) {
    const int64_t total = (int64_t)T * cout * cinp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cinp);
        const int64_t k = i / cinp;
        const int co = (int)(k % cout);
        const int t = (int)(k / cout);
        wb[i] = ci < cin ? w[((int64_t)co * cin + ci) * T + t] : 0.f;
    }
}

__global__ void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mfull, RowScale rs,
                                 const float* __restrict__ wf, const float* __restrict__ bias,
                                 const float* __restrict__ denom, const float* __restrict__ keep,
                                 ConvGeom g, int coutp, float* __restrict__ y) {
    const int G = coutp / 4;
    const int64_t total = (int64_t)g.n * g.ho * g.wo * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % G) * 4;
        const int64_t pix = idx / G;
        const int ox = (int)(pix % g.wo);
        const int oy = (int)((pix / g.wo) % g.ho);
        const int64_t n = pix / ((int64_t)g.wo * g.ho);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;
        if (kp) {
            for (int ky = 0; ky < g.kh; ++ky) {
                const int iy = oy * g.sh - g.ph + ky * g.dh;
                if (iy < 0 || iy >= g.h) continue;
                for (int kx = 0; kx < g.kw; ++kx) {
                    const int ix = ox * g.sw - g.pw + kx * g.dw;
                    if (ix < 0 || ix >= g.w) continue;
                    const int64_t ipix = (n * g.h + iy) * g.w + ix;
                    const float* xp = x + ipix * g.cin;
                    const float* wp = wf + (int64_t)(ky * g.kw + kx) * g.cin * coutp + co;
                    for (int ci = 0; ci < g.cin; ++ci) {
                        const float xv = xp[ci] * in_mask(mfull, rs, ipix, g.cin, ci);
                        const float4 wv = *reinterpret_cast<const float4*>(wp + (int64_t)ci * coutp);
                        a0 = fmaf(xv, wv.x, a0); a1 = fmaf(xv, wv.y, a1);
                        a2 = fmaf(xv, wv.z, a2); a3 = fmaf(xv, wv.w, a3);
                    }
                }
            }
        }
        const float dn = denom != nullptr ? denom[pix] : 1.f;
        float out[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (co + e >= g.cout) break;
            float v = out[e];
            if (kp) {
                if (denom != nullptr) v = v / dn;
                if (bias != nullptr) v += bias[co + e];
            } else {
                v = 0.f;
            }
            y[pix * g.cout + co + e] = v;
        }
    }
}

__global__ void dense_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                    const float* __restrict__ wb, const float* __restrict__ mfull, RowScale rs,
                                    ConvGeom g, int cinp, float* __restrict__ dx) {
    const int G = cinp / 4;
    const int64_t total = (int64_t)g.n * g.h * g.w * G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % G) * 4;
        const int64_t pix = idx / G;
        const int ix = (int)(pix % g.w);
        const int iy = (int)((pix / g.w) % g.h);
        const int64_t n = pix / ((int64_t)g.w * g.h);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
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
                if (s == 0.f) continue;
                const float* gp = dy + opix * g.cout;
                const float* wp = wb + (int64_t)(ky * g.kw + kx) * g.cout * cinp + ci;
                for (int co = 0; co < g.cout; ++co) {
                    const float gv = gp[co] * s;
                    const float4 wv = *reinterpret_cast<const float4*>(wp + (int64_t)co * cinp);
                    a0 = fmaf(gv, wv.x, a0); a1 = fmaf(gv, wv.y, a1);
                    a2 = fmaf(gv, wv.z, a2); a3 = fmaf(gv, wv.w, a3);
                }
            }
        }
        float out[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ci + e >= g.cin) break;
            dx[pix * g.cin + ci + e] = out[e] * in_mask(mfull, rs, pix, g.cin, ci + e);
        }
    }
}

// dW partials.  Block (kblock, coblock, chunk): thread = one im2col column k = (tap, ci), 32 output
// channels in registers; the block walks its pixel chunk in batches of 32 whose dy*inv rows and
// pixel coordinates are staged in LDS.  part[chunk][co][ci][t] (reference weight layout).
static constexpr int DD_CO = 32;
static constexpr int DD_PB = 32;
__global__ __launch_bounds__(256) void dense_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                           const float* __restrict__ x, const float* __restrict__ mfull,
                                                           RowScale rs, ConvGeom g, int64_t chunk,
                                                           float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float gs[DD_PB * DD_CO];
    __shared__ int pn[DD_PB], py[DD_PB], px[DD_PB];
    const int T = g.kh * g.kw;
    const int KK = T * g.cin;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * DD_CO;
    const int64_t npix = (int64_t)g.n * g.ho * g.wo;
    const int64_t pbeg = (int64_t)blockIdx.z * chunk;
    const int64_t pend = (pbeg + chunk < npix) ? pbeg + chunk : npix;
    const bool active = k < KK;
    const int t = active ? k / g.cin : 0;
    const int ci = active ? k % g.cin : 0;
    const int ky = t / g.kw, kx = t % g.kw;

    float acc[DD_CO];
#pragma unroll
    for (int j = 0; j < DD_CO; ++j) acc[j] = 0.f;

    for (int64_t p0 = pbeg; p0 < pend; p0 += DD_PB) {
        __syncthreads();
        for (int e = threadIdx.x; e < DD_PB * DD_CO; e += 256) {
            const int pp = e / DD_CO, j = e % DD_CO;
            const int64_t pix = p0 + pp;
            float v = 0.f;
            if (pix < pend && co0 + j < g.cout) {
                v = dy[pix * g.cout + co0 + j];
                if (inv != nullptr) v *= inv[pix];
            }
            gs[e] = v;
        }
        if (threadIdx.x < DD_PB) {
            const int64_t pix = p0 + threadIdx.x;
            if (pix < pend) {
                px[threadIdx.x] = (int)(pix % g.wo);
                py[threadIdx.x] = (int)((pix / g.wo) % g.ho);
                pn[threadIdx.x] = (int)(pix / ((int64_t)g.wo * g.ho));
            } else {
                pn[threadIdx.x] = -1;
            }
        }
        __syncthreads();
        if (active) {
            for (int pp = 0; pp < DD_PB; ++pp) {
                const int n = pn[pp];
                if (n < 0) break;
                const int iy = py[pp] * g.sh - g.ph + ky * g.dh;
                const int ix = px[pp] * g.sw - g.pw + kx * g.dw;
                if (iy < 0 || iy >= g.h || ix < 0 || ix >= g.w) continue;
                const int64_t ipix = ((int64_t)n * g.h + iy) * g.w + ix;
                const float xv = x[ipix * g.cin + ci] * in_mask(mfull, rs, ipix, g.cin, ci);
#pragma unroll
                for (int j = 0; j < DD_CO; ++j) acc[j] = fmaf(xv, gs[pp * DD_CO + j], acc[j]);
            }
        }
    }
    if (active) {
        float* pz = part + (int64_t)blockIdx.z * g.cout * g.cin * T;
#pragma unroll
        for (int j = 0; j < DD_CO; ++j)
            if (co0 + j < g.cout) pz[((int64_t)(co0 + j) * g.cin + ci) * T + t] = acc[j];
    }
}

// ---- few-output-channel path (Cout <= 4; ImageFill's final 35->3 layer) ---------------------
// A block owns an 8 x 32 tile of output pixels.  The masked input patch is staged once into LDS
// with fully coalesced row reads (NHWC rows are contiguous), so the 140-byte pixel stride of the
// 35-channel tensor never reaches the memory pipeline; each thread then walks its kh*kw*Cin
// window out of LDS (lane stride sw*Cin floats: conflict-free for odd Cin) against wave-uniform
// weights.
static constexpr int DS_TH = 8, DS_LDS = 12288;   // 48 KB patch budget -> 3 blocks per CU; tile width 32 or 16

struct SmallPlan {
    bool ok;
    int PH, PW, TW;
};
static SmallPlan plan_small(const ConvGeom& g) {
    SmallPlan p;
    p.PH = (DS_TH - 1) * g.sh + (g.kh - 1) * g.dh + 1;
    p.ok = false;
    for (int tw = 32; tw >= 16 && !p.ok; tw /= 2) {
        p.TW = tw;
        p.PW = (tw - 1) * g.sw + (g.kw - 1) * g.dw + 1;
        p.ok = g.cout <= 4 && (int64_t)p.PH * p.PW * g.cin <= DS_LDS && g.kh * g.kw * g.cin <= 1024;
    }
    return p;
}

// w[co][ci][t] -> w4[(t*cin + ci)*4 + co], zero padded to 4 output channels
__global__ void dense_prep_small_kernel(const float* __restrict__ w, int cin, int cout, int T, float* __restrict__ w4) {
    const int total = T * cin * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i & 3;
        const int k = i >> 2;
        const int ci = k % cin, t = k / cin;
        w4[i] = co < cout ? w[((int64_t)co * cin + ci) * T + t] : 0.f;
    }
}

// 8 patch rows at a time x 32 lanes striding one row: (px, ci) advance incrementally (no divisions) and the loads of
// a lane are independent, so many are in flight
__device__ __forceinline__ void stage_patch(float* __restrict__ tile, const float* __restrict__ x,
                                            const float* __restrict__ mfull, const RowScale& rs, const ConvGeom& g,
                                            int64_t n, int iy0, int ix0, int PH, int PW) {
    const int rowlen = PW * g.cin;
    const int lane = threadIdx.x & 31, prow = threadIdx.x >> 5;
    const int dpx = 32 / g.cin, dci = 32 % g.cin;
    for (int py = prow; py < PH; py += 8) {
        const int iy = iy0 + py;
        const bool yin = (iy >= 0 && iy < g.h);
        int px = lane / g.cin, ci = lane % g.cin;
        float* trow = tile + py * rowlen;
        for (int e = lane; e < rowlen; e += 32) {
            const int ix = ix0 + px;
            float v = 0.f;
            if (yin && ix >= 0 && ix < g.w) {
                const int64_t ipix = (n * g.h + iy) * g.w + ix;
                v = x[ipix * g.cin + ci] * in_mask(mfull, rs, ipix, g.cin, ci);
            }
            trow[e] = v;
            px += dpx; ci += dci;
            if (ci >= g.cin) { ci -= g.cin; ++px; }
        }
    }
}

template <int DS_TW>
__global__ __launch_bounds__(256) void dense_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mfull,
                                                              RowScale rs, const float* __restrict__ w4,
                                                              const float* __restrict__ bias, const float* __restrict__ denom,
                                                              const float* __restrict__ keep, ConvGeom g, int PH, int PW,
                                                              float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tile[DS_LDS];
    const int64_t n = blockIdx.z;
    const int oy0 = blockIdx.y * DS_TH, ox0 = blockIdx.x * DS_TW;
    stage_patch(tile, x, mfull, rs, g, n, oy0 * g.sh - g.ph, ox0 * g.sw - g.pw, PH, PW);
    __syncthreads();
    const int tx = threadIdx.x % DS_TW, ty = threadIdx.x / DS_TW;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (ty >= DS_TH || oy >= g.ho || ox >= g.wo) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ky = 0; ky < g.kh; ++ky)
        for (int kx = 0; kx < g.kw; ++kx) {
            const float* tp = tile + ((ty * g.sh + ky * g.dh) * PW + tx * g.sw + kx * g.dw) * g.cin;
            const float* wp = w4 + (ky * g.kw + kx) * g.cin * 4;
            for (int ci = 0; ci < g.cin; ++ci) {
                const float xv = tp[ci];
                a0 = fmaf(xv, wp[ci * 4 + 0], a0); a1 = fmaf(xv, wp[ci * 4 + 1], a1);
                a2 = fmaf(xv, wp[ci * 4 + 2], a2); a3 = fmaf(xv, wp[ci * 4 + 3], a3);
            }
        }
    const int64_t pix = (n * g.ho + oy) * g.wo + ox;
    const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;
    const float dn = denom != nullptr ? denom[pix] : 1.f;
    const float out[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e >= g.cout) break;
        float v = out[e];
        if (kp) {
            if (denom != nullptr) v = v / dn;
            if (bias != nullptr) v += bias[e];
        } else {
            v = 0.f;
        }
        y[pix * g.cout + e] = v;
    }
}

// dW partials for the few-output-channel path: persistent blocks walk tiles; thread t owns im2col
// columns t, t+256, ... (column = (tap, ci)) x 4 output channels in registers.
static constexpr int DS_NP = 4;
template <int DS_TW>
__global__ __launch_bounds__(256) void dense_small_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                                 const float* __restrict__ x, const float* __restrict__ mfull,
                                                                 RowScale rs, ConvGeom g, int PH, int PW, int tiles_per_block,
                                                                 float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float tile[DS_LDS];
    __shared__ __attribute__((aligned(16))) float gs[DS_TH * 32 * 4];
    const int T = g.kh * g.kw, KK = T * g.cin;
    const int ntx = (g.wo + DS_TW - 1) / DS_TW, nty = (g.ho + DS_TH - 1) / DS_TH;
    const int total_tiles = g.n * nty * ntx;
    int toff[DS_NP], tci[DS_NP];
#pragma unroll
    for (int q = 0; q < DS_NP; ++q) {
        const int k = threadIdx.x + 256 * q;
        const int t = k < KK ? k / g.cin : 0;
        tci[q] = k < KK ? k % g.cin : 0;
        toff[q] = ((t / g.kw) * g.dh * PW + (t % g.kw) * g.dw) * g.cin + tci[q];
    }
    float acc[DS_NP][4];
#pragma unroll
    for (int q = 0; q < DS_NP; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;

    const int t_beg = blockIdx.x * tiles_per_block;
    const int t_end = t_beg + tiles_per_block < total_tiles ? t_beg + tiles_per_block : total_tiles;
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int bx = tl % ntx, by = (tl / ntx) % nty;
        const int64_t n = tl / (ntx * nty);
        const int oy0 = by * DS_TH, ox0 = bx * DS_TW;
        __syncthreads();
        stage_patch(tile, x, mfull, rs, g, n, oy0 * g.sh - g.ph, ox0 * g.sw - g.pw, PH, PW);
        {
            const int tx = threadIdx.x % DS_TW, ty = threadIdx.x / DS_TW;
            const int oy = oy0 + ty, ox = ox0 + tx;
            float gv[4] = {0.f, 0.f, 0.f, 0.f};
            if (ty < DS_TH && oy < g.ho && ox < g.wo) {
                const int64_t pix = (n * g.ho + oy) * g.wo + ox;
                const float sc = inv != nullptr ? inv[pix] : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < g.cout) gv[e] = dy[pix * g.cout + e] * sc;
            }
            if (ty < DS_TH) *reinterpret_cast<float4*>(gs + threadIdx.x * 4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        __syncthreads();
        for (int py = 0; py < DS_TH; ++py)
            for (int px = 0; px < DS_TW; ++px) {
                const float4 gq = *reinterpret_cast<const float4*>(gs + (py * DS_TW + px) * 4);
                const int base = (py * g.sh * PW + px * g.sw) * g.cin;
#pragma unroll
                for (int q = 0; q < DS_NP; ++q) {
                    if (threadIdx.x + 256 * q >= KK) break;
                    const float xv = tile[base + toff[q]];
                    acc[q][0] = fmaf(xv, gq.x, acc[q][0]); acc[q][1] = fmaf(xv, gq.y, acc[q][1]);
                    acc[q][2] = fmaf(xv, gq.z, acc[q][2]); acc[q][3] = fmaf(xv, gq.w, acc[q][3]);
                }
            }
    }
    float* pz = part + (int64_t)blockIdx.x * g.cout * g.cin * T;
#pragma unroll
    for (int q = 0; q < DS_NP; ++q) {
        const int k = threadIdx.x + 256 * q;
        if (k >= KK) break;
        const int t = k / g.cin;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < g.cout) pz[((int64_t)e * g.cin + tci[q]) * T + t] = acc[q][e];
    }
}

static int small_dw_blocks(const ConvGeom& g, int tw, int* tiles_per_block) {
    const int ntx = cdiv(g.wo, tw), nty = cdiv(g.ho, DS_TH);
    const int total = g.n * nty * ntx;
    int blocks = total < 1024 ? total : 1024;
    *tiles_per_block = cdiv(total, blocks);
    return cdiv(total, *tiles_per_block);
}

// weight re-layouts for the implicit-GEMM path: w[co][ci][t] -> wr[co][t*cin + ci]  /  wd[ci][t*cout + co]
__global__ void conv_w_layout_kernel(const float* __restrict__ w, int cin, int cout, int T, int dx_layout, float* __restrict__ out) {
    const int64_t total = (int64_t)cout * cin * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const int ci = (int)((i / T) % cin);
        const int64_t co = i / ((int64_t)T * cin);
        const int64_t dst = dx_layout ? ((int64_t)ci * T + t) * cout + co : (co * T + t) * cin + ci;
        out[dst] = w[i];
    }
}

// fwd / dW: vector gather when the channels allow it, element-wise gather for stems / per-channel masks
static bool use_conv_gemm(const ConvGeom& g, const float* mfull, const void* a, const void* b, const void* c) {
    const ConvGemmGeom cg = {g.n, g.h, g.w, g.cin, g.cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw, g.ho, g.wo};
    if (!(aligned16(a) && aligned16(b) && aligned16(c))) return false;
    if (mfull == nullptr && conv_gemm_ok(cg)) return true;
    return conv_gemm_elem_ok(cg);
}
static bool use_conv_gemm_dx(const ConvGeom& g, const float* mfull, const void* a, const void* b, const void* c) {
    const ConvGemmGeom cg = {g.n, g.h, g.w, g.cin, g.cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw, g.ho, g.wo};
    return mfull == nullptr && conv_gemm_ok(cg) && g.cin >= 16 && aligned16(a) && aligned16(b) && aligned16(c);
}

static int check_conv_geom(const ConvGeom& g, const char* who) {
    TSII_REQUIRE(g.n > 0 && g.h > 0 && g.w > 0 && g.cin > 0 && g.cout > 0 && g.kh > 0 && g.kw > 0 && g.sh > 0 &&
                 g.sw > 0 && g.dh > 0 && g.dw > 0 && g.ph >= 0 && g.pw >= 0, "%s: bad geometry", who);
    TSII_REQUIRE(g.ho == (g.h + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1 &&
                 g.wo == (g.w + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1,
                 "%s: output size %dx%d inconsistent with conv geometry", who, g.ho, g.wo);
    return 0;
}

static inline int pad4(int v) { return (v + 3) / 4 * 4; }

struct DdPlan {
    int kblocks, coblocks, chunks;
    int64_t chunk;
};
static DdPlan plan_dd(const ConvGeom& g) {
    DdPlan p;
    const int KK = g.kh * g.kw * g.cin;
    p.kblocks = cdiv(KK, 256);
    p.coblocks = cdiv(g.cout, DD_CO);
    const int64_t npix = (int64_t)g.n * g.ho * g.wo;
    int64_t want = 1024 / ((int64_t)p.kblocks * p.coblocks);
    if (want < 1) want = 1;
    int64_t chunk = cdiv64(npix, want);
    if (chunk < 256) chunk = 256;
    chunk = cdiv64(chunk, DD_PB) * DD_PB;
    p.chunk = chunk;
    p.chunks = (int)cdiv64(npix, chunk);
    return p;
}

}  // namespace tsii

using namespace tsii;

#define CONV_GEOM() ConvGeom g = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo}

extern "C" size_t tsii_dense_ws_bytes(int cin, int cout, int kh, int kw) {
    if (cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0) return 0;
    const size_t a = (size_t)kh * kw * cin * (pad4(cout) < 4 ? 4 : pad4(cout));
    const size_t b = (size_t)kh * kw * cout * pad4(cin);
    return (a > b ? a : b) * sizeof(float);
}

extern "C" int tsii_dense_fwd(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                              const float* w, const float* bias, const float* denom, const float* keep,
                              int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && w && y && ws, "dense_fwd: null pointer");
    CONV_GEOM();
    if (check_conv_geom(g, "dense_fwd")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dense_ws_bytes(cin, cout, kh, kw), "dense_fwd: workspace too small");
    TSII_REQUIRE(aligned16(ws), "dense_fwd: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int coutp = pad4(cout), T = kh * kw;
    float* wf = (float*)ws;
    const SmallPlan sp = plan_small(g);
    if (sp.ok && cdiv(ho, DS_TH) <= 65535 && n <= 65535) {
        hipLaunchKernelGGL(dense_prep_small_kernel, dim3(cdiv(T * cin * 4, 256)), dim3(256), 0, st, w, cin, cout, T, wf);
        int rc0 = check_launch("dense_prep_small");
        if (rc0) return rc0;
        RowScale rs0 = {r0, r1, split};
        if (sp.TW == 32) hipLaunchKernelGGL((dense_small_fwd_kernel<32>), dim3(cdiv(wo, 32), cdiv(ho, DS_TH), n), dim3(256), 0, st, x, mfull,
                                            rs0, wf, bias, denom, keep, g, sp.PH, sp.PW, y);
        else hipLaunchKernelGGL((dense_small_fwd_kernel<16>), dim3(cdiv(wo, 16), cdiv(ho, DS_TH), n), dim3(256), 0, st, x, mfull,
                                rs0, wf, bias, denom, keep, g, sp.PH, sp.PW, y);
        return check_launch("dense_small_fwd");
    }
    if (use_conv_gemm(g, mfull, x, y, ws)) {   // MFMA implicit GEMM
        hipLaunchKernelGGL(conv_w_layout_kernel, dim3(stream_grid((int64_t)T * cin * cout, 256)), dim3(256), 0, st, w, cin, cout, T, 0, wf);
        int rcg = check_launch("conv_w_layout");
        if (rcg) return rcg;
        const ConvGemmGeom cgg = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
        RowScale rsg = {r0, r1, split};
        return launch_conv_gemm_fwd(x, mfull, rsg, wf, bias, denom, keep, cgg, y, st);
    }
    hipLaunchKernelGGL(dense_prep_fwd_kernel, dim3(stream_grid((int64_t)T * cin * coutp, 256)), dim3(256), 0, st,
                       w, cin, cout, T, coutp, wf);
    int rc = check_launch("dense_prep_fwd");
    if (rc) return rc;
    RowScale rs = {r0, r1, split};
    const int64_t total = (int64_t)n * ho * wo * (coutp / 4);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, x, mfull, rs, wf, bias,
                       denom, keep, g, coutp, y);
    return check_launch("dense_fwd");
}

extern "C" int tsii_dense_bwd_dx(const float* dy, const float* inv, const float* w, const float* mfull,
                                 const float* r0, int split, const float* r1, int n, int h, int wd, int cin, int cout,
                                 int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                 float* dx, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && w && dx && ws, "dense_bwd_dx: null pointer");
    CONV_GEOM();
    if (check_conv_geom(g, "dense_bwd_dx")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dense_ws_bytes(cin, cout, kh, kw), "dense_bwd_dx: workspace too small");
    TSII_REQUIRE(aligned16(ws), "dense_bwd_dx: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int cinp = pad4(cin), T = kh * kw;
    float* wb = (float*)ws;
    if (use_conv_gemm_dx(g, mfull, dy, dx, ws)) {
        hipLaunchKernelGGL(conv_w_layout_kernel, dim3(stream_grid((int64_t)T * cin * cout, 256)), dim3(256), 0, st, w, cin, cout, T, 1, wb);
        int rcg = check_launch("conv_w_layout");
        if (rcg) return rcg;
        const ConvGemmGeom cgg = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
        RowScale rsg = {r0, r1, split};
        return launch_conv_gemm_dx(dy, inv, wb, rsg, cgg, dx, st);
    }
    hipLaunchKernelGGL(dense_prep_dx_kernel, dim3(stream_grid((int64_t)T * cout * cinp, 256)), dim3(256), 0, st,
                       w, cin, cout, T, cinp, wb);
    int rc = check_launch("dense_prep_dx");
    if (rc) return rc;
    RowScale rs = {r0, r1, split};
    const int64_t total = (int64_t)n * h * wd * (cinp / 4);
    hipLaunchKernelGGL(dense_bwd_dx_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, dy, inv, wb, mfull, rs,
                       g, cinp, dx);
    return check_launch("dense_bwd_dx");
}

extern "C" size_t tsii_dense_bwd_dw_ws_bytes(int n, int ho, int wo, int cin, int cout, int kh, int kw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0) return 0;
    ConvGeom g = {n, 0, 0, cin, cout, kh, kw, 1, 1, 0, 0, 1, 1, ho, wo};
    DdPlan p = plan_dd(g);
    int tpb = 0;
    const int small_rows = small_dw_blocks(g, 16, &tpb);   // upper bound over both tile widths
    const int rows = p.chunks > small_rows ? p.chunks : small_rows;
    size_t main_floats = (size_t)rows * cout * cin * kh * kw;
    const ConvGemmGeom cgg = {n, 0, 0, cin, cout, kh, kw, 1, 1, 0, 0, 1, 1, ho, wo};
    if ((conv_gemm_ok(cgg) || conv_gemm_elem_ok(cgg)) && conv_gemm_dw_ws_floats(cgg) > main_floats) main_floats = conv_gemm_dw_ws_floats(cgg);
    return (main_floats + colsum_ws_floats((int64_t)n * ho * wo, cout)) * sizeof(float);
}

extern "C" int tsii_dense_bwd_dw(const float* dy, const float* inv, const float* keep, const float* x, const float* mfull,
                                 const float* r0, int split, const float* r1, int n, int h, int wd, int cin, int cout,
                                 int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                 float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && x && dwgt && ws, "dense_bwd_dw: null pointer");
    CONV_GEOM();
    if (check_conv_geom(g, "dense_bwd_dw")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, kh, kw), "dense_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DdPlan p = plan_dd(g);
    RowScale rs = {r0, r1, split};
    float* part = (float*)ws;
    if (use_conv_gemm(g, mfull, dy, x, ws)) {
        const ConvGemmGeom cgg = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
        int rcg = launch_conv_gemm_dw(dy, inv, x, mfull, rs, cgg, dwgt, part, st);
        if (rcg) return rcg;
        if (dbias != nullptr)
            rcg = launch_colsum_scaled(dy, keep, (int64_t)n * ho * wo, cout, dbias, part + conv_gemm_dw_ws_floats(cgg), st);
        return rcg;
    }
    const SmallPlan sp = plan_small(g);
    if (sp.ok) {
        int tpb = 0;
        const int blocks = small_dw_blocks(g, sp.TW, &tpb);
        if (sp.TW == 32) hipLaunchKernelGGL((dense_small_bwd_dw_kernel<32>), dim3(blocks), dim3(256), 0, st, dy, inv, x, mfull, rs, g, sp.PH,
                                            sp.PW, tpb, part);
        else hipLaunchKernelGGL((dense_small_bwd_dw_kernel<16>), dim3(blocks), dim3(256), 0, st, dy, inv, x, mfull, rs, g, sp.PH,
                                sp.PW, tpb, part);
        int rc1 = check_launch("dense_small_bwd_dw");
        if (rc1) return rc1;
        const int64_t len1 = (int64_t)cout * cin * kh * kw;
        rc1 = launch_reduce_rows(part, blocks, len1, dwgt, st);
        if (rc1) return rc1;
        if (dbias != nullptr)
            rc1 = launch_colsum_scaled(dy, keep, (int64_t)n * ho * wo, cout, dbias, part + (size_t)blocks * len1, st);
        return rc1;
    }
    hipLaunchKernelGGL(dense_bwd_dw_kernel, dim3(p.kblocks, p.coblocks, p.chunks), dim3(256), 0, st, dy, inv, x, mfull,
                       rs, g, p.chunk, part);
    int rc = check_launch("dense_bwd_dw");
    if (rc) return rc;
    const int64_t len = (int64_t)cout * cin * kh * kw;
    rc = launch_reduce_rows(part, p.chunks, len, dwgt, st);
    if (rc) return rc;
    if (dbias != nullptr)
        rc = launch_colsum_scaled(dy, keep, (int64_t)n * ho * wo, cout, dbias, part + (size_t)p.chunks * len, st);
    return rc;
}
