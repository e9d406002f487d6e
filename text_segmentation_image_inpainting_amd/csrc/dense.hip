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

// ---- head path: 3x3 / stride 1 / dilation 1, Cout <= 4, plane masks --------------------------------------------
// ImageFill's 35->3 output layer (and the 67->3 heads of ImageFillOrigin / V2) at full resolution: 8.4 M pixels x
// 945 MACs -- 0.3 ms of HBM traffic, but the generic few-output-channel kernels above spend ~10 ms on it per
// step: their element-wise staging waits on every load and the odd pixel stride (35 floats) forces scalar LDS
// reads.  Here the patch is staged with 8 independent loads in flight per thread into pixels padded to CP = 4*CG
// floats (36-float stride: the conflict-free ds_read_b128 pattern of the GEMM tiles), the x*mask multiply rides in
// the staging (mask planes staged first), weights are wave-uniform (scalar loads), and all loops are unrolled.
template <int CG>
struct Head {
    static constexpr int CP = CG * 4;
    static constexpr int TH = 8;
    static constexpr int TW = (10 * 34 * CP <= 12288) ? 32 : 16;
    static constexpr int NPX = TH * TW;
    static constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
    static constexpr int TILE = NPIX * CP;
    static constexpr int SIDE = (2 * NPIX > NPX * 4) ? 2 * NPIX : NPX * 4;   // mask planes of the patch, later dy*inv
    static_assert(TILE <= 12288, "patch must fit 48 KB");
};

// exact e / d for 0 <= e < 2^24 without an integer division
__device__ __forceinline__ int fdiv_small(int e, int d, float inv_d) {
    int q = (int)((float)e * inv_d);
    if (q * d > e) --q;
    else if ((q + 1) * d <= e) ++q;
    return q;
}

// K4c: the head's input as a VIRTUAL concatenation cat(nearest-x2(low [n,h/2,w/2,c1]), skip [n,h,w,c2]) -- the decoder's last
// DoubleUpSample + torch.cat (models/image_inpainting.py:82-85) is never written: the patch is staged straight from the two
// tensors (16-byte loads for the c1 % 4 == 0 up-sampled channels, 4x fewer bytes than the concatenated tensor), the mask split
// is the concat boundary.  low == nullptr: plain input x.
struct HeadCat {
    const float* low;
    const float* skip;
    int c1, c2;
};

template <int CG>
__device__ __forceinline__ void head_stage(float* __restrict__ tile, float* __restrict__ side, const float* __restrict__ x,
                                           const HeadCat& cat, const RowScale& rs, const ConvGeom& g, int64_t n, int iy0, int ix0) {
    using H = Head<CG>;
    const int tid = threadIdx.x, nt = blockDim.x;
    // 1. the two mask planes of the patch; pad channels of every pixel are zero
    for (int p = tid; p < H::NPIX; p += nt) {
        const int py = p / H::PW, px = p - py * H::PW;
        const int iy = iy0 + py, ix = ix0 + px;
        float m0 = 0.f, m1 = 0.f;
        if (iy >= 0 && iy < g.h && ix >= 0 && ix < g.w) {
            const int64_t ipix = (n * g.h + iy) * g.w + ix;
            m0 = rs.r0 != nullptr ? rs.r0[ipix] : 1.f;
            m1 = rs.r1 != nullptr ? rs.r1[ipix] : 1.f;          // the two planes are independent (NULL = ones)
        }
        side[p] = m0;
        side[H::NPIX + p] = m1;
        for (int c = g.cin; c < H::CP; ++c) tile[p * H::CP + c] = 0.f;
    }
    __syncthreads();
    if (cat.low != nullptr) {
        // 2a. up-sampled part: element e = (patch pixel p, channel quad q) <- low pixel (iy >> 1, ix >> 1), times plane 0
        const int q1 = cat.c1 >> 2, h2 = g.h >> 1, w2 = g.w >> 1;
        const float inv_q1 = 1.0f / (float)q1;
        const int tot1 = H::NPIX * q1;
        for (int e0 = tid; e0 < tot1; e0 += nt * 4) {
            float4 v[4];
            int dst[4], pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * nt;
                const int p = fdiv_small(e, q1, inv_q1);
                const int q = e - p * q1;
                const int py = p / H::PW, px = p - py * H::PW;
                const int iy = iy0 + py, ix = ix0 + px;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[u] = p * H::CP + q * 4;
                pp[u] = p;
                if (e < tot1 && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w)
                    v[u] = *reinterpret_cast<const float4*>(cat.low + ((n * h2 + (iy >> 1)) * w2 + (ix >> 1)) * cat.c1 + q * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * nt < tot1) {
                    const float m = side[pp[u]];
                    *reinterpret_cast<float4*>(tile + dst[u]) = make_float4(v[u].x * m, v[u].y * m, v[u].z * m, v[u].w * m);
                }
        }
        // 2b. skip part: element e = (patch pixel p, channel ci of c2), times plane 1
        const int tot2 = H::NPIX * cat.c2;
        const float inv_c2 = 1.0f / (float)cat.c2;
        for (int e0 = tid; e0 < tot2; e0 += nt * 4) {
            float v[4];
            int dst[4], pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * nt;
                const int p = fdiv_small(e, cat.c2, inv_c2);
                const int ci = e - p * cat.c2;
                const int py = p / H::PW, px = p - py * H::PW;
                const int iy = iy0 + py, ix = ix0 + px;
                v[u] = 0.f;
                dst[u] = p * H::CP + cat.c1 + ci;
                pp[u] = p;
                if (e < tot2 && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w)
                    v[u] = cat.skip[((n * g.h + iy) * g.w + ix) * cat.c2 + ci];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * nt < tot2) tile[dst[u]] = v[u] * side[H::NPIX + pp[u]];
        }
        return;
    }
    // 2. x * mask: element e = (patch pixel p, channel ci); a patch row is one contiguous run of the NHWC tensor
    const int total = H::NPIX * g.cin;
    const float inv_cin = 1.0f / (float)g.cin;
    const int split = rs.r0 != nullptr ? rs.split : 0x7fffffff;
    for (int e0 = tid; e0 < total; e0 += nt * 8) {
        float v[8];
        int dst[8], msk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * nt;
            const int p = fdiv_small(e, g.cin, inv_cin);
            const int ci = e - p * g.cin;
            const int py = p / H::PW, px = p - py * H::PW;
            const int iy = iy0 + py, ix = ix0 + px;
            v[u] = 0.f;
            dst[u] = p * H::CP + ci;
            msk[u] = (ci < split ? 0 : H::NPIX) + p;
            if (e < total && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w)
                v[u] = x[((n * g.h + iy) * g.w + ix) * g.cin + ci];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * nt < total) tile[dst[u]] = v[u] * side[msk[u]];
    }
}

// w[co][ci][t] -> w4[((t*CG + cg)*4 + j)*4 + co], ci = cg*4 + j, zero padded
__global__ void head_prep_fwd_kernel(const float* __restrict__ w, int cin, int cout, int CG, float* __restrict__ w4) {
    const int total = 9 * CG * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i & 3, j = (i >> 2) & 3;
        const int cg = (i >> 4) % CG, t = (i >> 4) / CG;
        const int ci = cg * 4 + j;
        w4[i] = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * 9 + t] : 0.f;
    }
}
// w[co][ci][t] -> wd[(t*4 + co)*CP + ci], zero padded
__global__ void head_prep_dx_kernel(const float* __restrict__ w, int cin, int cout, int CP, float* __restrict__ wd) {
    const int total = 9 * 4 * CP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % CP, co = (i / CP) & 3, t = i / (4 * CP);
        wd[i] = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * 9 + t] : 0.f;
    }
}

template <int CG, int NCO>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, HeadCat cat, RowScale rs, const float* __restrict__ w4,
                                                       const float* __restrict__ bias, const float* __restrict__ denom,
                                                       const float* __restrict__ keep, ConvGeom g, float* __restrict__ y) {
    using H = Head<CG>;
    __shared__ __attribute__((aligned(16))) float tile[H::TILE];
    __shared__ __attribute__((aligned(16))) float side[H::SIDE];
    const int64_t n = blockIdx.z;
    const int oy0 = blockIdx.y * H::TH, ox0 = blockIdx.x * H::TW;
    head_stage<CG>(tile, side, x, cat, rs, g, n, oy0 - g.ph, ox0 - g.pw);
    __syncthreads();
    constexpr int SPLIT = 256 / H::NPX;             // waves sharing a pixel (TW 16): each takes every SPLIT-th group
    const int pixl = threadIdx.x % H::NPX;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / H::NPX);
    const int tx = pixl % H::TW, ty = pixl / H::TW;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float* tp = tile + ((ty + t / 3) * H::PW + tx + t % 3) * H::CP;
#pragma unroll
        for (int j = 0; j < (CG + SPLIT - 1) / SPLIT; ++j) {
            const int cg = j * SPLIT + part;
            if (cg >= CG) break;
            const float4 xv = *reinterpret_cast<const float4*>(tp + cg * 4);
            const float* __restrict__ wp = w4 + (t * CG + cg) * 16;
#pragma unroll
            for (int e = 0; e < NCO; ++e) {
                a[e] = fmaf(xv.x, wp[e], a[e]); a[e] = fmaf(xv.y, wp[4 + e], a[e]);
                a[e] = fmaf(xv.z, wp[8 + e], a[e]); a[e] = fmaf(xv.w, wp[12 + e], a[e]);
            }
        }
    }
    if constexpr (SPLIT > 1) {
        __syncthreads();
        if (part != 0) *reinterpret_cast<float4*>(tile + ((part - 1) * H::NPX + pixl) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int q = 1; q < SPLIT; ++q) {
                const float4 o = *reinterpret_cast<const float4*>(tile + ((q - 1) * H::NPX + pixl) * 4);
                a[0] += o.x; a[1] += o.y; a[2] += o.z; a[3] += o.w;
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (part != 0 || oy >= g.ho || ox >= g.wo) return;
    const int64_t pix = (n * g.ho + oy) * g.wo + ox;
    const bool kp = keep != nullptr ? (keep[pix] != 0.f) : true;
    const float dn = denom != nullptr ? denom[pix] : 1.f;
#pragma unroll
    for (int e = 0; e < NCO; ++e) {
        if (e >= g.cout) break;
        float v = a[e];
        if (kp) {
            if (denom != nullptr) v = v / dn;
            if (bias != nullptr) v += bias[e];
        } else {
            v = 0.f;
        }
        y[pix * g.cout + e] = v;
    }
}

// dW partials: persistent block of 320 threads.  Thread (ci = tid % CP, strip = tid / CP) owns input channel ci and the
// output rows of its strip, with all 9 taps x NCO output channels (27 / 36 accumulators) in registers.  Per 8-pixel row
// segment it pulls a 3 x 10 window of x (30 ds_read_b32, consecutive lanes = consecutive channels: conflict free) and the
// 8 dy*inv pixels (8 ds_read_b128, broadcast within the strip) and issues 9 x 8 x NCO FMAs: ~6 FMAs per LDS read.
// (The first version gave every thread one im2col column and walked the pixels: 1 ds_read_b32 + 1 broadcast b128 per 3
// FMAs -- LDS-issue bound at 9 % of the HBM roofline, 2.4 ms for ImageFill's 35 -> 3 head.)
template <int CG, int NCO>
__global__ __launch_bounds__(320) void head_dw_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                      const float* __restrict__ x, HeadCat cat, RowScale rs, ConvGeom g,
                                                      int tiles_per_block, float* __restrict__ part) {
    using H = Head<CG>;
    __shared__ __attribute__((aligned(16))) float tile[H::TILE];
    __shared__ __attribute__((aligned(16))) float side[H::SIDE];
    constexpr int CP = H::CP, NS = 320 / CP, RPS = H::TH / NS, W = 8;
    static_assert(NS >= 1 && H::TH % NS == 0 && H::TW % W == 0, "strip geometry");
    static_assert(NS * 9 * NCO * CP <= H::TILE, "strip reduction must fit the patch LDS");
    const int ntx = (g.wo + H::TW - 1) / H::TW, nty = (g.ho + H::TH - 1) / H::TH;
    const int total_tiles = g.n * nty * ntx;
    const int ci = threadIdx.x % CP, strip = threadIdx.x / CP;
    const bool worker = strip < NS;
    float acc[9][NCO];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < NCO; ++e) acc[t][e] = 0.f;
    const int t_beg = blockIdx.x * tiles_per_block;
    const int t_end = t_beg + tiles_per_block < total_tiles ? t_beg + tiles_per_block : total_tiles;
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int bx = tl % ntx, by = (tl / ntx) % nty;
        const int64_t n = tl / (ntx * nty);
        const int oy0 = by * H::TH, ox0 = bx * H::TW;
        float gv[4] = {0.f, 0.f, 0.f, 0.f};           // this thread's dy*inv pixel, in flight during the staging
        if (threadIdx.x < H::NPX) {
            const int oy = oy0 + threadIdx.x / H::TW, ox = ox0 + threadIdx.x % H::TW;
            if (oy < g.ho && ox < g.wo) {
                const int64_t pix = (n * g.ho + oy) * g.wo + ox;
                const float sc = inv != nullptr ? inv[pix] : 1.f;
#pragma unroll
                for (int e = 0; e < NCO; ++e)
                    if (e < g.cout) gv[e] = dy[pix * g.cout + e] * sc;
            }
        }
        __syncthreads();                               // previous tile fully consumed
        head_stage<CG>(tile, side, x, cat, rs, g, n, oy0 - g.ph, ox0 - g.pw);
        __syncthreads();                               // patch ready, mask planes dead
        if (threadIdx.x < H::NPX) *reinterpret_cast<float4*>(side + threadIdx.x * 4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        __syncthreads();
        if (worker) {
#pragma unroll
            for (int r = 0; r < RPS; ++r) {
                const int py = strip * RPS + r;
#pragma unroll 1
                for (int px0 = 0; px0 < H::TW; px0 += W) {
                    float xr[3][W + 2];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int j = 0; j < W + 2; ++j) xr[ky][j] = tile[((py + ky) * H::PW + px0 + j) * CP + ci];
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float4 gq = *reinterpret_cast<const float4*>(side + (py * H::TW + px0 + j) * 4);
                        const float gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                                for (int e = 0; e < NCO; ++e) acc[ky * 3 + kx][e] = fmaf(xr[ky][j + kx], gg[e], acc[ky * 3 + kx][e]);
                    }
                }
            }
        }
    }
    // strips -> one partial per block: red[(strip * 9*NCO + t*NCO + e) * CP + ci]
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < NCO; ++e) tile[((strip * 9 + t) * NCO + e) * CP + ci] = acc[t][e];
    }
    __syncthreads();
    float* pz = part + (int64_t)blockIdx.x * g.cout * g.cin * 9;
    for (int k = threadIdx.x; k < 9 * NCO * CP; k += 320) {
        const int c = k % CP, e = (k / CP) % NCO, t = k / (CP * NCO);
        if (c >= g.cin || e >= g.cout) continue;
        float a = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) a += tile[((sidx * 9 + t) * NCO + e) * CP + c];
        pz[((int64_t)e * g.cin + c) * 9 + t] = a;
    }
}

// dX: thread = one input pixel x up to 9 channel groups (36 accumulators); dy*inv patch in LDS, weights wave-uniform;
// the 35-float pixels leave through an LDS transpose so that the global stores are contiguous runs
template <int CG, int NCO>
__global__ __launch_bounds__(256) void head_dx_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                      const float* __restrict__ wd, RowScale rs, ConvGeom g,
                                                      float* __restrict__ dx, float* __restrict__ dlow, float* __restrict__ dskip, int c1) {
    using H = Head<CG>;
    constexpr int SPLIT = 256 / H::NPX;
    constexpr int CGT = (CG + SPLIT - 1) / SPLIT;     // channel groups per thread
    __shared__ __attribute__((aligned(16))) float outt[H::NPX * H::CP];
    __shared__ __attribute__((aligned(16))) float gp[H::NPIX * 4];
    __shared__ float mk[2 * H::NPX];
    const int64_t n = blockIdx.z;
    const int iy0 = blockIdx.y * H::TH, ix0 = blockIdx.x * H::TW;
    // dy*inv patch: patch pixel r,c <-> output pixel (iy0 + ph - 2 + r, ix0 + pw - 2 + c)
    for (int p = threadIdx.x; p < H::NPIX; p += 256) {
        const int r = p / H::PW, c = p - r * H::PW;
        const int oy = iy0 + g.ph - 2 + r, ox = ix0 + g.pw - 2 + c;
        float gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (oy >= 0 && oy < g.ho && ox >= 0 && ox < g.wo) {
            const int64_t pix = (n * g.ho + oy) * g.wo + ox;
            const float sc = inv != nullptr ? inv[pix] : 1.f;
#pragma unroll
            for (int e = 0; e < NCO; ++e)
                if (e < g.cout) gv[e] = dy[pix * g.cout + e] * sc;
        }
        *reinterpret_cast<float4*>(gp + p * 4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
    for (int p = threadIdx.x; p < H::NPX; p += 256) {   // x*mask multiplied back in (partial_convolution.py:51)
        const int iy = iy0 + p / H::TW, ix = ix0 + p % H::TW;
        float m0 = 0.f, m1 = 0.f;
        if (iy < g.h && ix < g.w) {
            const int64_t ipix = (n * g.h + iy) * g.w + ix;
            m0 = rs.r0 != nullptr ? rs.r0[ipix] : 1.f;
            m1 = rs.r1 != nullptr ? rs.r1[ipix] : 1.f;          // the two planes are independent (NULL = ones)
        }
        mk[p] = m0; mk[H::NPX + p] = m1;
    }
    __syncthreads();
    const int pixl = threadIdx.x % H::NPX;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / H::NPX);
    const int tx = pixl % H::TW, ty = pixl / H::TW;
    const int cg0 = part * CGT;
    float acc[CGT][4];
#pragma unroll
    for (int j = 0; j < CGT; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 gq = *reinterpret_cast<const float4*>(gp + ((ty + 2 - t / 3) * H::PW + tx + 2 - t % 3) * 4);
        const float gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
        for (int j = 0; j < CGT; ++j) {
            if (cg0 + j >= CG) break;
#pragma unroll
            for (int e = 0; e < NCO; ++e) {
                const float* __restrict__ wp = wd + (t * 4 + e) * H::CP + (cg0 + j) * 4;
                acc[j][0] = fmaf(gg[e], wp[0], acc[j][0]); acc[j][1] = fmaf(gg[e], wp[1], acc[j][1]);
                acc[j][2] = fmaf(gg[e], wp[2], acc[j][2]); acc[j][3] = fmaf(gg[e], wp[3], acc[j][3]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CGT; ++j)
        if (cg0 + j < CG)
            *reinterpret_cast<float4*>(outt + pixl * H::CP + (cg0 + j) * 4) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    __syncthreads();
    if (dx == nullptr) {
        // K4c: the gradient of the virtual concatenation leaves as its two pieces -- d skip [n,h,w,c2] (if wanted) and
        // d low [n,h/2,w/2,c1] = the sum over the 2x2 pixels a low-resolution pixel was copied to (tiles start on even rows / columns)
        const int c2 = g.cin - c1;
        if (dskip != nullptr) {
            for (int e = threadIdx.x; e < H::NPX * c2; e += 256) {
                const int p = e / c2, ci = e - p * c2;
                const int iy = iy0 + p / H::TW, ix = ix0 + p % H::TW;
                if (iy < g.h && ix < g.w) dskip[((n * g.h + iy) * g.w + ix) * c2 + ci] = outt[p * H::CP + c1 + ci] * mk[H::NPX + p];
            }
        }
        const int h2 = g.h >> 1, w2 = g.w >> 1, q1 = c1 >> 2;
        constexpr int LW = H::TW / 2, LH = H::TH / 2;
        for (int e = threadIdx.x; e < LH * LW * q1; e += 256) {
            const int q = e % q1, lp = e / q1;
            const int lx = lp % LW, ly = lp / LW;
            const int gy = (iy0 >> 1) + ly, gx = (ix0 >> 1) + lx;
            if (gy >= h2 || gx >= w2) continue;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int p = (2 * ly + (d >> 1)) * H::TW + 2 * lx + (d & 1);
                const float4 v = *reinterpret_cast<const float4*>(outt + p * H::CP + q * 4);
                const float m = mk[p];
                a.x = fmaf(v.x, m, a.x); a.y = fmaf(v.y, m, a.y); a.z = fmaf(v.z, m, a.z); a.w = fmaf(v.w, m, a.w);
            }
            *reinterpret_cast<float4*>(dlow + ((n * h2 + gy) * w2 + gx) * c1 + q * 4) = a;
        }
        return;
    }
    // write-out: tile row ty = one contiguous run of min(TW, w - ix0) * cin floats
    const int vw = g.w - ix0 < H::TW ? g.w - ix0 : H::TW;
    const int run = vw * g.cin;
    const float inv_cin = 1.0f / (float)g.cin;
    const int split = rs.r0 != nullptr ? rs.split : 0x7fffffff;
    for (int r = 0; r < H::TH; ++r) {
        const int iy = iy0 + r;
        if (iy >= g.h) break;
        float* drow = dx + ((n * g.h + iy) * g.w + ix0) * g.cin;
        for (int e = threadIdx.x; e < run; e += 256) {
            const int px = fdiv_small(e, g.cin, inv_cin);
            const int ci = e - px * g.cin;
            const int p = r * H::TW + px;
            drow[e] = outt[p * H::CP + ci] * mk[(ci < split ? 0 : H::NPX) + p];
        }
    }
}

static int head_cg(const ConvGeom& g, const float* mfull) {   // channel groups of the head instantiation, 0 = not applicable
    if (!(g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.dh == 1 && g.dw == 1 && g.cout <= 4 && mfull == nullptr)) return 0;
    if (cdiv(g.ho, 8) > 65535 || cdiv(g.h, 8) > 65535 || g.n > 65535) return 0;
    if (g.cin > 20 && g.cin <= 36) return 9;
    if (g.cin > 36 && g.cin <= 68) return 17;
    return 0;
}
static int head_dw_blocks(const ConvGeom& g, int tw, int* tiles_per_block) {
    const int total = g.n * cdiv(g.ho, 8) * cdiv(g.wo, tw);
    const int blocks = total < 768 ? total : 768;     // 3 resident blocks per CU
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

// the three head launches, shared by the dense entry points (plain input x) and the K4c ones (virtual concatenation)
static int launch_head_fwd(const float* x, const HeadCat& cat, const RowScale& rsh, const float* w, const float* bias, const float* denom,
                           const float* keep, const ConvGeom& g, int hcg, float* wf, float* y, hipStream_t st) {
    hipLaunchKernelGGL(head_prep_fwd_kernel, dim3(cdiv(9 * hcg * 16, 256)), dim3(256), 0, st, w, g.cin, g.cout, hcg, wf);
    int rch = check_launch("head_prep_fwd");
    if (rch) return rch;
    const dim3 grid(cdiv(g.wo, hcg == 9 ? 32 : 16), cdiv(g.ho, 8), g.n);
    if (hcg == 9 && g.cout <= 3) hipLaunchKernelGGL((head_fwd_kernel<9, 3>), grid, dim3(256), 0, st, x, cat, rsh, wf, bias, denom, keep, g, y);
    else if (hcg == 9) hipLaunchKernelGGL((head_fwd_kernel<9, 4>), grid, dim3(256), 0, st, x, cat, rsh, wf, bias, denom, keep, g, y);
    else if (g.cout <= 3) hipLaunchKernelGGL((head_fwd_kernel<17, 3>), grid, dim3(256), 0, st, x, cat, rsh, wf, bias, denom, keep, g, y);
    else hipLaunchKernelGGL((head_fwd_kernel<17, 4>), grid, dim3(256), 0, st, x, cat, rsh, wf, bias, denom, keep, g, y);
    return check_launch("head_fwd");
}
static int launch_head_dx(const float* dy, const float* inv, const float* w, const RowScale& rsh, const ConvGeom& g, int hcg, float* wb,
                          float* dx, float* dlow, float* dskip, int c1, hipStream_t st) {
    hipLaunchKernelGGL(head_prep_dx_kernel, dim3(cdiv(9 * 4 * hcg * 4, 256)), dim3(256), 0, st, w, g.cin, g.cout, hcg * 4, wb);
    int rch = check_launch("head_prep_dx");
    if (rch) return rch;
    const dim3 grid(cdiv(g.w, hcg == 9 ? 32 : 16), cdiv(g.h, 8), g.n);
    if (hcg == 9 && g.cout <= 3) hipLaunchKernelGGL((head_dx_kernel<9, 3>), grid, dim3(256), 0, st, dy, inv, wb, rsh, g, dx, dlow, dskip, c1);
    else if (hcg == 9) hipLaunchKernelGGL((head_dx_kernel<9, 4>), grid, dim3(256), 0, st, dy, inv, wb, rsh, g, dx, dlow, dskip, c1);
    else if (g.cout <= 3) hipLaunchKernelGGL((head_dx_kernel<17, 3>), grid, dim3(256), 0, st, dy, inv, wb, rsh, g, dx, dlow, dskip, c1);
    else hipLaunchKernelGGL((head_dx_kernel<17, 4>), grid, dim3(256), 0, st, dy, inv, wb, rsh, g, dx, dlow, dskip, c1);
    return check_launch("head_dx");
}
static int launch_head_dw(const float* dy, const float* inv, const float* keep, const float* x, const HeadCat& cat, const RowScale& rs,
                          const ConvGeom& g, int hcg, float* part, float* dwgt, float* dbias, hipStream_t st) {
    int tpb = 0;
    const int blocks = head_dw_blocks(g, hcg == 9 ? 32 : 16, &tpb);
    if (hcg == 9 && g.cout <= 3) hipLaunchKernelGGL((head_dw_kernel<9, 3>), dim3(blocks), dim3(320), 0, st, dy, inv, x, cat, rs, g, tpb, part);
    else if (hcg == 9) hipLaunchKernelGGL((head_dw_kernel<9, 4>), dim3(blocks), dim3(320), 0, st, dy, inv, x, cat, rs, g, tpb, part);
    else if (g.cout <= 3) hipLaunchKernelGGL((head_dw_kernel<17, 3>), dim3(blocks), dim3(320), 0, st, dy, inv, x, cat, rs, g, tpb, part);
    else hipLaunchKernelGGL((head_dw_kernel<17, 4>), dim3(blocks), dim3(320), 0, st, dy, inv, x, cat, rs, g, tpb, part);
    int rch = check_launch("head_dw");
    if (rch) return rch;
    const int64_t lenh = (int64_t)g.cout * g.cin * 9;
    rch = launch_reduce_rows(part, blocks, lenh, dwgt, st);
    if (rch) return rch;
    if (dbias != nullptr)
        rch = launch_colsum_scaled(dy, keep, (int64_t)g.n * g.ho * g.wo, g.cout, dbias, part + (size_t)blocks * lenh, st);
    return rch;
}
static const HeadCat kNoCat = {nullptr, nullptr, 0, 0};

}  // namespace tsii

using namespace tsii;

#define CONV_GEOM() ConvGeom g = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo}

extern "C" size_t tsii_dense_ws_bytes(int cin, int cout, int kh, int kw) {
    if (cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0) return 0;
    size_t fl = (size_t)kh * kw * pad4(cin) * (pad4(cout) < 4 ? 4 : pad4(cout));
    if (kh == 3 && kw == 3 && cout <= 4 && cin <= 68) {   // head kernels pad the channels to their instantiation (36 / 68)
        const size_t head = (size_t)9 * (cin <= 36 ? 36 : 68) * 4;
        if (head > fl) fl = head;
    }
    return fl * sizeof(float);
}

static int dense_fwd_impl(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                          const float* w, const float* bias, const float* denom, const float* keep,
                          int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw,
                          int dh, int dw, int ho, int wo, float* stats, float* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && w && y && ws, "dense_fwd: null pointer");
    CONV_GEOM();
    if (check_conv_geom(g, "dense_fwd")) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_dense_ws_bytes(cin, cout, kh, kw), "dense_fwd: workspace too small");
    TSII_REQUIRE(aligned16(ws), "dense_fwd: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int coutp = pad4(cout), T = kh * kw;
    float* wf = (float*)ws;
    TSII_REQUIRE(stats == nullptr || (head_cg(g, mfull) == 0 && !plan_small(g).ok && use_conv_gemm(g, mfull, x, y, ws)),
                 "dense_fwd_bn: statistics partials need the implicit-GEMM path (tsii_dense_stat_rows() > 0, aligned operands)");
    if (const int hcg = head_cg(g, mfull)) {       // 3x3 few-output-channel head
        const RowScale rsh = {r0, r1, split};
        return launch_head_fwd(x, kNoCat, rsh, w, bias, denom, keep, g, hcg, wf, y, st);
    }
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
    // Forward only: the gather needs Cin % 4 == 0 but nothing of Cout, so convs with very few output channels over many
    // input channels (the 128 -> 1 logits conv of the segmentation nets: 8.5 ms per step on the generic kernel) take the
    // 128x32 GEMM tile too -- 31/32 of its MFMA columns idle, still several times faster.
    const bool few_out_gemm = stats == nullptr && mfull == nullptr && cin % 4 == 0 && cout < 16 && T * cin >= 128 &&
                              aligned16(x) && aligned16(ws);
    if (use_conv_gemm(g, mfull, x, y, ws) || few_out_gemm) {   // MFMA implicit GEMM
        hipLaunchKernelGGL(conv_w_layout_kernel, dim3(stream_grid((int64_t)T * cin * cout, 256)), dim3(256), 0, st, w, cin, cout, T, 0, wf);
        int rcg = check_launch("conv_w_layout");
        if (rcg) return rcg;
        const ConvGemmGeom cgg = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
        RowScale rsg = {r0, r1, split};
        return launch_conv_gemm_fwd(x, mfull, rsg, wf, bias, denom, keep, cgg, y, st, stats);
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

extern "C" int tsii_dense_fwd(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                              const float* w, const float* bias, const float* denom, const float* keep,
                              int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw,
                              int dh, int dw, int ho, int wo, float* y, void* ws, size_t ws_bytes, void* stream) {
    return dense_fwd_impl(x, mfull, r0, split, r1, w, bias, denom, keep, n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo,
                          nullptr, y, ws, ws_bytes, stream);
}

extern "C" int64_t tsii_dense_stat_rows(int has_mfull, int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw,
                                        int ph, int pw, int dh, int dw, int ho, int wo) {
    if (n <= 0 || ho <= 0 || wo <= 0) return 0;
    const ConvGeom g = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
    const ConvGemmGeom cg = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
    if (head_cg(g, has_mfull ? (const float*)1 : nullptr) != 0 || plan_small(g).ok) return 0;
    if (!((has_mfull == 0 && conv_gemm_ok(cg)) || conv_gemm_elem_ok(cg))) return 0;
    return cdiv64((int64_t)n * ho * wo, 128);
}

extern "C" int tsii_dense_fwd_bn(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                                 const float* w, const float* bias, const float* denom, const float* keep,
                                 int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw,
                                 int dh, int dw, int ho, int wo, float* stat_part, float* y, void* ws, size_t ws_bytes,
                                 void* stream) {
    TSII_REQUIRE(stat_part != nullptr, "dense_fwd_bn: null stat_part");
    return dense_fwd_impl(x, mfull, r0, split, r1, w, bias, denom, keep, n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo,
                          stat_part, y, ws, ws_bytes, stream);
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
    if (const int hcg = head_cg(g, mfull)) {       // 3x3 few-output-channel head
        const RowScale rsh = {r0, r1, split};
        return launch_head_dx(dy, inv, w, rsh, g, hcg, wb, dx, nullptr, nullptr, 0, st);
    }
    if (use_conv_gemm_dx(g, mfull, dy, dx, ws)) {
        const ConvGemmGeom cgp = {n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo};
        if (conv_gemm_dx_phases_ok(cgp)) {           // strided: one stride-1 problem per stride phase
            RowScale rsp = {r0, r1, split};
            return launch_conv_gemm_dx_phases(dy, inv, w, wb, rsp, cgp, dx, st);
        }
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
    if (const int hcg = head_cg(g, mfull)) {       // 3x3 few-output-channel head
        return launch_head_dw(dy, inv, keep, x, kNoCat, rs, g, hcg, part, dwgt, dbias, st);
    }
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


// ---- K4c: 3x3 / stride 1 / pad 1 head over the virtual concatenation cat(nearest-x2(low), skip) -----------------------------
// (models/image_inpainting.py:82-86: DoubleUpSample + torch.cat + the 35 -> 3 output PartialConv; here the concatenated tensor and
// its gradient never exist).  r0 / r1: the mask planes of the two parts at full resolution (NULL = all ones); the split is c1.
extern "C" int tsii_head_cat_ok(int n, int h, int wd, int c1, int c2, int cout) {
    if (n <= 0 || h <= 0 || wd <= 0 || c1 <= 0 || c2 <= 0 || cout <= 0 || (h & 1) || (wd & 1) || (c1 & 3)) return 0;
    const ConvGeom g = {n, h, wd, c1 + c2, cout, 3, 3, 1, 1, 1, 1, 1, 1, h, wd};
    return head_cg(g, nullptr) != 0 ? 1 : 0;
}

extern "C" int tsii_head_cat_fwd(const float* low, const float* skip, int c1, int c2, const float* r0, const float* r1,
                                 const float* w, const float* bias, const float* denom, const float* keep,
                                 int n, int h, int wd, int cout, float* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(low && skip && w && y && ws, "head_cat_fwd: null pointer");
    TSII_REQUIRE(tsii_head_cat_ok(n, h, wd, c1, c2, cout), "head_cat_fwd: geometry has no fused head (tsii_head_cat_ok)");
    TSII_REQUIRE(aligned16(low) && aligned16(ws) && ws_bytes >= tsii_dense_ws_bytes(c1 + c2, cout, 3, 3), "head_cat_fwd: workspace / alignment");
    const ConvGeom g = {n, h, wd, c1 + c2, cout, 3, 3, 1, 1, 1, 1, 1, 1, h, wd};
    const HeadCat cat = {low, skip, c1, c2};
    const RowScale rsh = {r0, r1, c1};
    return launch_head_fwd(nullptr, cat, rsh, w, bias, denom, keep, g, head_cg(g, nullptr), (float*)ws, y, (hipStream_t)stream);
}

extern "C" int tsii_head_cat_bwd_dx(const float* dy, const float* inv, const float* w, int c1, int c2, const float* r0, const float* r1,
                                    int n, int h, int wd, int cout, float* dlow, float* dskip, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && w && dlow && ws, "head_cat_bwd_dx: null pointer");
    TSII_REQUIRE(tsii_head_cat_ok(n, h, wd, c1, c2, cout), "head_cat_bwd_dx: geometry has no fused head (tsii_head_cat_ok)");
    TSII_REQUIRE(aligned16(dlow) && aligned16(ws) && ws_bytes >= tsii_dense_ws_bytes(c1 + c2, cout, 3, 3), "head_cat_bwd_dx: workspace / alignment");
    const ConvGeom g = {n, h, wd, c1 + c2, cout, 3, 3, 1, 1, 1, 1, 1, 1, h, wd};
    const RowScale rsh = {r0, r1, c1};
    return launch_head_dx(dy, inv, w, rsh, g, head_cg(g, nullptr), (float*)ws, nullptr, dlow, dskip, c1, (hipStream_t)stream);
}

extern "C" int tsii_head_cat_bwd_dw(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                    int c1, int c2, const float* r0, const float* r1, int n, int h, int wd, int cout,
                                    float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && low && skip && dwgt && ws, "head_cat_bwd_dw: null pointer");
    TSII_REQUIRE(tsii_head_cat_ok(n, h, wd, c1, c2, cout), "head_cat_bwd_dw: geometry has no fused head (tsii_head_cat_ok)");
    TSII_REQUIRE(aligned16(low) && ws_bytes >= tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3), "head_cat_bwd_dw: workspace / alignment");
    const ConvGeom g = {n, h, wd, c1 + c2, cout, 3, 3, 1, 1, 1, 1, 1, 1, h, wd};
    const HeadCat cat = {low, skip, c1, c2};
    const RowScale rs = {r0, r1, c1};
    return launch_head_dw(dy, inv, keep, nullptr, cat, rs, g, head_cg(g, nullptr), (float*)ws, dwgt, dbias, (hipStream_t)stream);
}
