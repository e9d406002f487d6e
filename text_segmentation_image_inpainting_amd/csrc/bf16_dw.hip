// bf16 ACTIVATION STORAGE, depth-wise 3x3 convolutions and average pools of the segmentation nets
// (Conv_block(groups = C) of models/BaseModels.py:105-127 as used by models/Xception.py:13-44 and models/MobileNetV2.py:136-138;
// nn.AvgPool2d(k, 1, (k-1)//2) of models/common.py:62-68): forward, dX, dW on bf16 NHWC tensors, fp32 arithmetic.
//
// HBM-bound stencils.  With 8 channels per 16-byte vector a thread can afford to keep its own window: a thread owns ONE
// output column x and ONE channel octet and MARCHES down the rows of its chunk with a ring of the 3 input rows x 3 column
// taps it needs, as the packed bf16 it loaded (36 VGPRs); every step it requests the next step's new row (3 loads, or 6 at
// stride 2) before it computes, so a row is fetched once per thread and the x-neighbours' overlapping taps (three threads
// read every input pixel) are L1 / L2 hits of loads issued by the same wave in the same step.  No LDS tile, no halo
// bookkeeping, any dilation: with dilation d the rows of one residue class mod d form an ordinary 3-row stencil, so a
// thread walks ONE row phase (rows r, r+d, r+2d ...) and the phases are independent blocks.  The 9 x 8 weights of an octet
// (and the load-time BatchNorm constants, K6b) sit in LDS and are read where they are used (broadcast reads): the kernel
// stays under 128 VGPRs = 4 waves per SIMD, which is what hides the HBM latency of the one-step-ahead requests.
//   hdw_conv_kernel   forward (EPI 1: BatchNorm statistics partials of the rounded output, K6b) and -- with flipped taps and
//                     padding 2d - p -- the stride-1 dX (EPI 2: the BatchNorm-backward reductions of the producer, K6c)
//   hdw_dw_kernel     weight gradient: 72 accumulators per thread, block reduction through LDS, partial rows + row reduce
//   hdw_dx_s2_kernel  dX of the two stride-2 layers (gather form)
//   havgpool_kernel   k x k / stride 1 / count_include_pad average pool (its own adjoint): ring of k horizontal sums
#include "bf16_common.h"
#include "dw_lean_api.h"

namespace tsii {

struct HDwPlan {
    int n, hin, win, c, hout, wout;   // input tensor [n, hin, win, c] -> output [n, hout, wout, c]
    int s, p, d, flip;                // stride, padding, dilation; flip: use taps (2-ky, 2-kx) (the adjoint)
    int cg, pxb, ncg, colgroups, phases, chunks, jper;
    int nx;                           // output columns per thread: 2 at stride 1 (x and x + d share two of their three tap columns), else 1
};

// nxmode: 0 one output column per thread; 1 two where the geometry allows (stride 1, pxb % d == 0); 2 the same for d <= 2 only.
// Measured on cfg 5's layers (profiles/r05aa_dw_nx.log): forward + K6b 94-100 -> 82 us, plain dX 56-65 -> 54-55, dX + K6c 117-136 -> 113-123
// at d <= 2 but 117 -> 122 at d = 4, weight gradient 104-110 -> 111-118 (231-256 VGPRs): so forward / plain dX 1, K6c 2, weight gradient 0
static HDwPlan hdw_plan(int n, int hin, int win, int c, int hout, int wout, int s, int p, int d, int flip, int nxmode) {
    HDwPlan g;
    g.n = n; g.hin = hin; g.win = win; g.c = c; g.hout = hout; g.wout = wout; g.s = s; g.p = p; g.d = d; g.flip = flip;
    const int oct = c / 8;
    int cg = 1;
    while (cg * 2 <= oct && cg * 2 <= 16) cg *= 2;
    g.cg = cg; g.pxb = 256 / cg;
    g.ncg = cdiv(oct, cg);
#ifndef HDW_NX2
#define HDW_NX2 1        // 0: one output column per thread everywhere (A/B, tools/variants)
#endif
    g.nx = (HDW_NX2 && nxmode != 0 && (nxmode == 1 || d <= 2) && s == 1 && g.pxb % d == 0 && wout >= 2 * d) ? 2 : 1;
    g.colgroups = cdiv(wout, g.pxb * g.nx);
    g.phases = s == 1 ? (d < hout ? d : hout) : 1;
    const int hp = cdiv(hout, g.phases);                       // rows of the longest phase
    const int64_t base = (int64_t)n * g.colgroups * g.ncg * g.phases;
    int64_t chunks = cdiv64(2048, base);
    const int maxc = hp / 8 > 1 ? hp / 8 : 1;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    g.jper = cdiv(hp, (int)chunks);
    g.chunks = cdiv(hp, g.jper);
    return g;
}
static inline int64_t hdw_rows(const HDwPlan& g) { return (int64_t)g.n * g.phases * g.chunks * g.colgroups; }
static inline int64_t hdw_blocks(const HDwPlan& g) { return hdw_rows(g) * g.ncg; }

struct HDwBn {          // K6c: BatchNorm whose backward reductions the dX kernel takes (raw input bn_y at the output positions)
    const bf16_t* y;
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    float eps, neg, hi;
};

// block coordinates from the (XCD-remapped) block id: column group fastest, then channel group, chunk, phase, image
struct HDwBlock {
    int colg, cgi, chunk, phase, n;
    int64_t prow;
};
__device__ __forceinline__ HDwBlock hdw_decode(const HDwPlan& g) {
    unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    HDwBlock k;
    k.colg = (int)(b % g.colgroups); b /= g.colgroups;
    k.cgi = (int)(b % g.ncg); b /= g.ncg;
    k.chunk = (int)(b % g.chunks); b /= g.chunks;
    k.phase = (int)(b % g.phases); b /= g.phases;
    k.n = (int)b;
    k.prow = (((int64_t)k.n * g.phases + k.phase) * g.chunks + k.chunk) * g.colgroups + k.colg;
    return k;
}

// what a marching thread needs to address its window: NC = 2 + (output columns per thread) input columns, d apart
template <int NC>
struct HDwWin {
    const bf16_t* src;     // image base + channel offset
    int64_t rowstride;     // win * c
    int c;
    int xc[NC];            // clamped input columns
    unsigned xmask;        // bit kc: column inside the image
    int b, dq, hin;        // input row of sequence index q: b + dq * q
};
// this thread's first output column inside the block's window of pxb * nx columns: with two columns per thread (x and x + d) the
// threads px = 0 .. pxb - 1 take x = 2 d (px / d) + px % d, so that the pairs tile the window
__device__ __forceinline__ int hdw_first_col(const HDwPlan& g, int colg, int px) {
    return colg * g.pxb * g.nx + (g.nx == 2 ? 2 * g.d * (px / g.d) + px % g.d : px);
}
template <int NC>
__device__ __forceinline__ void hdw_window(HDwWin<NC>& wn, const HDwPlan& g, int xo, int S) {
    wn.xmask = 0u;
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) {
        const int xi = xo * S - g.p + kc * g.d;
        const bool v = xi >= 0 && xi < g.win;
        wn.xc[kc] = v ? xi : 0;
        wn.xmask |= v ? (1u << kc) : 0u;
    }
}
template <int NC>
__device__ __forceinline__ void hdw_load_row(const HDwWin<NC>& wn, int q, hu32x4 (&dst)[NC], unsigned& ok) {
    const int iy = wn.b + wn.dq * q;
    const bool rv = iy >= 0 && iy < wn.hin;
    const int iyc = iy < 0 ? 0 : (iy >= wn.hin ? wn.hin - 1 : iy);
    const bf16_t* rp = wn.src + (int64_t)iyc * wn.rowstride;
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) dst[kc] = ld8(rp + (int64_t)wn.xc[kc] * wn.c);
    ok = rv ? wn.xmask : 0u;
}
// the producer's BatchNorm + activation (the virtual activation is a bf16 tensor: rounded like a stored one), zero padding
template <bool BNIN, int NC>
__device__ __forceinline__ void hdw_commit(hu32x4 (&r)[NC], unsigned ok, const float* __restrict__ lsc, const float* __restrict__ lsh, float neg, float hi) {
#pragma unroll
    for (int kc = 0; kc < NC; ++kc) {
        if constexpr (BNIN) {
            float v[8];
            unpack8(r[kc], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bn_act_load(v[e], lsc[e], lsh[e], neg, hi);
            r[kc] = pack8(v);
        }
        const hu32x4 z = {0u, 0u, 0u, 0u};
        r[kc] = ((ok >> kc) & 1u) ? r[kc] : z;
    }
}

// LDS of the marching kernels: [0, 9*128) weights [tap][octet*8+e]; then 4 x 128 constants; then the reduction scratch
static constexpr int HDW_W = 0, HDW_C0 = 9 * 128, HDW_RED = HDW_C0 + 4 * 128;

template <int S, bool BNIN, int EPI, int NX>
__global__ __launch_bounds__(256) void hdw_conv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       HDwPlan g, InBN ib, float* __restrict__ part, HDwBn kb, bf16_t* __restrict__ y) {
    static_assert(NX == 1 || S == 1, "two output columns per thread at stride 1 only");
    constexpr int NC = NX + 2;
    __shared__ __attribute__((aligned(16))) float sm[HDW_RED + 256 * 16 + 128];
    const HDwBlock bk = hdw_decode(g);
    const int tid = threadIdx.x;
    const int o = tid % g.cg, px = tid / g.cg;
    const int oct = bk.cgi * g.cg + o;
    const bool ch_ok = oct * 8 < g.c;
    const int c0 = ch_ok ? oct * 8 : 0;
    const int xo = hdw_first_col(g, bk.colg, px);               // output columns xo (+ i d, i < NX)
    const bool x_ok = xo < g.wout;
    // weights (flipped for the adjoint) and per-channel constants of this block's channel group
    for (int i = tid; i < 9 * g.cg * 8; i += 256) {
        const int t = i / (g.cg * 8), ce = i % (g.cg * 8);
        const int ch = bk.cgi * g.cg * 8 + ce;
        sm[HDW_W + t * 128 + ce] = ch < g.c ? w[(int64_t)ch * 9 + (g.flip ? 8 - t : t)] : 0.f;
    }
    for (int i = tid; i < g.cg * 8; i += 256) {
        const int ch = bk.cgi * g.cg * 8 + i;
        const bool okc = ch < g.c;
        if constexpr (BNIN) { sm[HDW_C0 + i] = okc ? ib.sc[ch] : 0.f; sm[HDW_C0 + 128 + i] = okc ? ib.sh[ch] : 0.f; }
        if constexpr (EPI == 2) {
            sm[HDW_C0 + i] = okc ? kb.mean[ch] : 0.f; sm[HDW_C0 + 128 + i] = okc ? 1.0f / sqrtf(kb.var[ch] + kb.eps) : 0.f;
            sm[HDW_C0 + 256 + i] = okc ? kb.gamma[ch] : 0.f; sm[HDW_C0 + 384 + i] = okc ? kb.beta[ch] : 0.f;
        }
    }
    __syncthreads();
    const float* lw = sm + HDW_W + o * 8;
    const float* lc = sm + HDW_C0 + o * 8;

    // rows of this block: output row of step j = (S == 1 ? phase + d j : j), j in [j0, j1)
    const int hp = S == 1 ? (g.hout - bk.phase + g.d - 1) / g.d : g.hout;
    const int j0 = bk.chunk * g.jper;
    int j1 = j0 + g.jper;
    j1 = j1 < hp ? j1 : hp;
    float s1[8], s2[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; pv[e] = 0.f; }
    const bool active = j0 < j1;           // (block-uniform) a late phase may have no rows in the last chunk
    if (active) {
        HDwWin<NC> wn;
        wn.src = x + (int64_t)bk.n * g.hin * g.win * g.c + c0;
        wn.rowstride = (int64_t)g.win * g.c;
        wn.c = g.c; wn.hin = g.hin;
        hdw_window<NC>(wn, g, x_ok ? xo : 0, S);
        wn.b = S == 1 ? bk.phase - g.p : -g.p;
        wn.dq = S == 1 ? g.d : 1;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias != nullptr && ch_ok) ? bias[c0 + e] : 0.f;
        const float neg = ib.neg, hi = ib.hi;

        hu32x4 ring[3][NC], nxt[S][NC], yq[NX];
        unsigned rok[3], nok[S];
#pragma unroll
        for (int k = 0; k < 3; ++k) hdw_load_row<NC>(wn, S * j0 + k, ring[k], rok[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) hdw_commit<BNIN, NC>(ring[k], rok[k], lc, lc + 128, neg, hi);
        const int64_t orow = (int64_t)g.wout * g.c;
        bf16_t* yout = y + (int64_t)bk.n * g.hout * orow + (int64_t)(x_ok ? xo : 0) * g.c + c0;
        const bf16_t* ybn = EPI == 2 ? kb.y + (int64_t)bk.n * g.hout * orow + (int64_t)(x_ok ? xo : 0) * g.c + c0 : nullptr;
        bool st_ok[NX];
        int64_t coff[NX];                  // element offset of output column i from column xo (clamped inside the row when it is outside)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            st_ok[i] = ch_ok && xo + i * g.d < g.wout;
            coff[i] = st_ok[i] ? (int64_t)i * g.d * g.c : 0;
        }
        for (int j = j0; j < j1; ++j) {
            const int yo = S == 1 ? bk.phase + g.d * j : j;
            // the next step's new rows (clamped rows past the end are loaded and never used)
#pragma unroll
            for (int i = 0; i < S; ++i) hdw_load_row<NC>(wn, S * (j + 1) + 3 - S + i, nxt[i], nok[i]);
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < NX; ++i) yq[i] = ld8(ybn + (int64_t)yo * orow + coff[i]);
            }
            float acc[NX][8];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][e] = bv[e];
            // every window vector is unpacked once and feeds the outputs it belongs to: column kc is tap kx = kc - i of output i
            // (the weights are read where they are used -- twice per step with two outputs: LDS broadcast reads, no registers held)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    float v[8];
                    unpack8(ring[k][kc], v);
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        if (kc - i < 0 || kc - i > 2) continue;
                        const int t = k * 3 + kc - i;
                        const float4 w0 = *reinterpret_cast<const float4*>(lw + t * 128), w1 = *reinterpret_cast<const float4*>(lw + t * 128 + 4);
                        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][e] = fmaf(wv[e], v[e], acc[i][e]);
                    }
                }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const hu32x4 ob = pack8(acc[i]);
                if (st_ok[i]) st8_nt(yout + (int64_t)yo * orow + coff[i], ob);
                if constexpr (EPI == 1) {
                    float v[8];
                    unpack8(ob, v);
                    if (i == 0 && j == j0) {      // block pivot per channel: the first output of the block's first column (block-uniform branch)
                        if (px == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) sm[HDW_RED + o * 8 + e] = v[e];
                        }
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < 8; ++e) pv[e] = sm[HDW_RED + o * 8 + e];
                        __syncthreads();
                    }
                    if (st_ok[i]) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float dlt = v[e] - pv[e];
                            s1[e] += dlt;
                            s2[e] = fmaf(dlt, dlt, s2[e]);
                        }
                    }
                }
                if constexpr (EPI == 2) {
                    if (st_ok[i]) {
                        float v[8], yv[8];
                        unpack8(ob, v);
                        unpack8(yq[i], yv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = (yv[e] - lc[e]) * lc[128 + e];
                            const float z = fmaf(xh, lc[256 + e], lc[384 + e]);
                            const float dz = v[e] * inbn_grad(z, kb.neg, kb.hi);
                            s1[e] += dz;
                            s2[e] = fmaf(dz, xh, s2[e]);
                        }
                    }
                }
            }
            // rotate the window
#pragma unroll
            for (int i = 0; i < S; ++i) hdw_commit<BNIN, NC>(nxt[i], nok[i], lc, lc + 128, neg, hi);
#pragma unroll
            for (int k = 0; k < 3 - S; ++k)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) ring[k][kc] = ring[k + S][kc];
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) ring[3 - S + i][kc] = nxt[i][kc];
        }
    }
    if constexpr (EPI != 0) {
        // block reduction over the pixels that share a channel: [256][16] scratch, then one partial row per block
        __syncthreads();
        float* red = sm + HDW_RED;
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
        if (EPI == 1 && px == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[HDW_RED + 256 * 16 + o * 8 + e] = pv[e];
        }
        __syncthreads();
        if (tid < g.cg * 8) {
            const int oo = tid >> 3, e = tid & 7;
            const int ch = (bk.cgi * g.cg + oo) * 8 + e;
            if (ch < g.c) {
                float a1 = 0.f, a2 = 0.f;
                for (int p = 0; p < g.pxb; ++p) { a1 += red[(p * g.cg + oo) * 16 + e]; a2 += red[(p * g.cg + oo) * 16 + 8 + e]; }
                if constexpr (EPI == 1) {
                    int vpx = g.wout - bk.colg * g.pxb * g.nx;
                    vpx = vpx < g.pxb * g.nx ? vpx : g.pxb * g.nx;
                    float* sp = part + bk.prow * 4 * g.c;
                    sp[ch] = active ? (float)((int64_t)vpx * (j1 - j0)) : 0.f;
                    sp[g.c + ch] = sm[HDW_RED + 256 * 16 + oo * 8 + e];
                    sp[2 * g.c + ch] = a1;
                    sp[3 * g.c + ch] = a2;
                } else {
                    float* sp = part + bk.prow * 2 * g.c;
                    sp[ch] = a1;
                    sp[g.c + ch] = a2;
                }
            }
        }
    }
}

// ---- weight gradient: dw[c][ky][kx] = sum dy[n,yo,xo,c] * a[n, yo*S - p + ky*d, xo*S - p + kx*d, c] ------------------------
template <int S, bool BNIN, int NX>
__global__ __launch_bounds__(256) void hdw_dw_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, HDwPlan g, InBN ib,
                                                     float* __restrict__ part) {
    static_assert(NX == 1 || S == 1, "two output columns per thread at stride 1 only");
    constexpr int NC = NX + 2;
    __shared__ __attribute__((aligned(16))) float sm[256 + 256 * 8];
    const HDwBlock bk = hdw_decode(g);
    const int tid = threadIdx.x;
    const int o = tid % g.cg, px = tid / g.cg;
    const int oct = bk.cgi * g.cg + o;
    const bool ch_ok = oct * 8 < g.c;
    const int c0 = ch_ok ? oct * 8 : 0;
    const int xo = hdw_first_col(g, bk.colg, px);
    const bool x_ok = xo < g.wout;
    for (int i = tid; i < g.cg * 8; i += 256) {
        const int ch = bk.cgi * g.cg * 8 + i;
        const bool okc = ch < g.c;
        sm[i] = (BNIN && okc) ? ib.sc[ch] : 0.f; sm[128 + i] = (BNIN && okc) ? ib.sh[ch] : 0.f;
    }
    __syncthreads();
    const float* lc = sm + o * 8;
    const int hp = S == 1 ? (g.hout - bk.phase + g.d - 1) / g.d : g.hout;
    const int j0 = bk.chunk * g.jper;
    int j1 = j0 + g.jper;
    j1 = j1 < hp ? j1 : hp;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    if (j0 < j1) {
        HDwWin<NC> wn;
        wn.src = x + (int64_t)bk.n * g.hin * g.win * g.c + c0;
        wn.rowstride = (int64_t)g.win * g.c;
        wn.c = g.c; wn.hin = g.hin;
        hdw_window<NC>(wn, g, x_ok ? xo : 0, S);
        wn.b = S == 1 ? bk.phase - g.p : -g.p;
        wn.dq = S == 1 ? g.d : 1;
        const float neg = ib.neg, hi = ib.hi;
        hu32x4 ring[3][NC], nxt[S][NC];
        unsigned rok[3], nok[S];
#pragma unroll
        for (int k = 0; k < 3; ++k) hdw_load_row<NC>(wn, S * j0 + k, ring[k], rok[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) hdw_commit<BNIN, NC>(ring[k], rok[k], lc, lc + 128, neg, hi);
        const int64_t orow = (int64_t)g.wout * g.c;
        const bf16_t* dyp = dy + (int64_t)bk.n * g.hout * orow + (int64_t)(x_ok ? xo : 0) * g.c + c0;
        bool st_ok[NX];
        int64_t coff[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            st_ok[i] = ch_ok && xo + i * g.d < g.wout;
            coff[i] = st_ok[i] ? (int64_t)i * g.d * g.c : 0;
        }
        for (int j = j0; j < j1; ++j) {
            const int yo = S == 1 ? bk.phase + g.d * j : j;
#pragma unroll
            for (int i = 0; i < S; ++i) hdw_load_row<NC>(wn, S * (j + 1) + 3 - S + i, nxt[i], nok[i]);
            float gv[NX][8];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                unpack8(ld8(dyp + (int64_t)yo * orow + coff[i]), gv[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[i][e] = st_ok[i] ? gv[i][e] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    float v[8];
                    unpack8(ring[k][kc], v);
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        if (kc - i < 0 || kc - i > 2) continue;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[k * 3 + kc - i][e] = fmaf(gv[i][e], v[e], acc[k * 3 + kc - i][e]);
                    }
                }
#pragma unroll
            for (int i = 0; i < S; ++i) hdw_commit<BNIN, NC>(nxt[i], nok[i], lc, lc + 128, neg, hi);
#pragma unroll
            for (int k = 0; k < 3 - S; ++k)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) ring[k][kc] = ring[k + S][kc];
#pragma unroll
            for (int i = 0; i < S; ++i)
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) ring[3 - S + i][kc] = nxt[i][kc];
        }
    }
    // per tap: the block's pixels are summed through LDS; partial row [c][9] per block
    float* red = sm + 256;
    float* sp = part + bk.prow * 9 * g.c;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[t][e];
        __syncthreads();
        if (tid < g.cg * 8) {
            const int oo = tid >> 3, e = tid & 7;
            const int ch = (bk.cgi * g.cg + oo) * 8 + e;
            if (ch < g.c) {
                float a = 0.f;
                for (int p = 0; p < g.pxb; ++p) a += red[(p * g.cg + oo) * 8 + e];
                sp[(int64_t)ch * 9 + t] = a;
            }
        }
    }
}

// ---- dX of a stride-2 layer (d = 1): dx[y][x] = sum over the taps with (y + p - ky, x + p - kx) both even: 1, 2 or 4 of the 9 --
// the others are skipped, not loaded-and-zeroed (a thread's taps depend on the parities of its pixel only).  Weights of all
// channels in LDS ([tap][c], c <= 1024).
static constexpr int HDW_S2_MAXC = 1024;
__global__ __launch_bounds__(256) void hdw_dx_s2_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ w, int n, int h, int wd, int c,
                                                        int ho, int wo, int p, bf16_t* __restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float lw[9 * HDW_S2_MAXC];
    for (int i = threadIdx.x; i < 9 * c; i += 256) { const int t = i / c, ch = i - t * c; lw[i] = w[(int64_t)ch * 9 + t]; }
    __syncthreads();
    const int G = c / 8;
    const int64_t total = (int64_t)n * h * wd * G;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c0 = (int)(idx % G) * 8;
    const int64_t pix = idx / G;
    const int x = (int)(pix % wd), yy = (int)((pix / wd) % h);
    const int64_t b = pix / ((int64_t)wd * h);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ky = (yy + p) & 1; ky < 3; ky += 2) {
        const int ty = yy + p - ky;
        if (ty < 0 || (ty >> 1) >= ho) continue;
        for (int kx = (x + p) & 1; kx < 3; kx += 2) {
            const int tx = x + p - kx;
            if (tx < 0 || (tx >> 1) >= wo) continue;
            float v[8];
            unpack8(ld8(dy + ((b * ho + (ty >> 1)) * wo + (tx >> 1)) * c + c0), v);
            const float* wt = lw + (ky * 3 + kx) * c + c0;
            const float4 w0 = *reinterpret_cast<const float4*>(wt), w1 = *reinterpret_cast<const float4*>(wt + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], wv[e], acc[e]);
        }
    }
    st8_nt(dx + pix * c + c0, pack8(acc));
}

// ---- k x k average pool, stride 1, padding (k-1)/2, count_include_pad (divisor always k^2): a thread marches down one column
// with a ring of the last K horizontal sums (fp32); the output row is their sum ------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void havgpool_kernel(const bf16_t* __restrict__ x, HDwPlan g, bf16_t* __restrict__ y) {
    constexpr int R = (K - 1) / 2;
    const HDwBlock bk = hdw_decode(g);
    const int tid = threadIdx.x;
    const int o = tid % g.cg, px = tid / g.cg;
    const int oct = bk.cgi * g.cg + o;
    const bool ch_ok = oct * 8 < g.c;
    const int c0 = ch_ok ? oct * 8 : 0;
    const int xo = bk.colg * g.pxb + px;
    const bool x_ok = xo < g.wout;
    const int j0 = bk.chunk * g.jper;
    int j1 = j0 + g.jper;
    j1 = j1 < g.hout ? j1 : g.hout;
    if (j0 >= j1) return;
    const bf16_t* src = x + (int64_t)bk.n * g.hin * g.win * g.c + c0;
    const int64_t rstride = (int64_t)g.win * g.c;
    const int xb = x_ok ? xo : 0;
    auto hsum = [&](int iy, float (&hs)[8]) {
        const bool rv = iy >= 0 && iy < g.hin;
        const int iyc = iy < 0 ? 0 : (iy >= g.hin ? g.hin - 1 : iy);
        hu32x4 t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int xi = xb - R + k;
            const int xc = xi < 0 ? 0 : (xi >= g.win ? g.win - 1 : xi);
            t[k] = ld8(src + (int64_t)iyc * rstride + (int64_t)xc * g.c);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) hs[e] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int xi = xb - R + k;
            const bool ok = rv && xi >= 0 && xi < g.win;
            float v[8];
            unpack8(t[k], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) hs[e] += ok ? v[e] : 0.f;
        }
    };
    float ring[K][8];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) hsum(j0 - R + k, ring[k + 1]);     // rows j0-R .. j0+R-1 sit in slots 1 .. K-1
    const float inv = 1.0f / (float)(K * K);
    bf16_t* yout = y + (int64_t)bk.n * g.hout * rstride + (int64_t)xb * g.c + c0;
    for (int j = j0; j < j1; ++j) {
#pragma unroll
        for (int k = 0; k < K - 1; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) ring[k][e] = ring[k + 1][e];
        hsum(j + R, ring[K - 1]);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) a += ring[k][e];
            acc[e] = a * inv;
        }
        if (x_ok && ch_ok) st8_nt(yout + (int64_t)j * rstride, pack8(acc));
    }
}

static int hdw_geom_ok(const char* who, int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo) {
    TSII_REQUIRE(n > 0 && h > 0 && wd > 0 && c > 0 && c % 8 == 0, "%s: bad shape (channels must be a multiple of 8, got %d)", who, c);
    TSII_REQUIRE(kh == 3 && kw == 3 && sh == sw && (sh == 1 || sh == 2) && dh == dw && dh >= 1 && ph == pw && ph >= 0 && (sh == 1 || dh == 1),
                 "%s: the bf16 depth-wise kernels are 3x3, stride 1 (any dilation) or stride 2 (dilation 1), square geometry", who);
    TSII_REQUIRE(ho == (h + 2 * ph - dh * 2 - 1) / sh + 1 && wo == (wd + 2 * pw - dw * 2 - 1) / sw + 1 && ho > 0 && wo > 0, "%s: output size does not match the geometry", who);
    TSII_REQUIRE((int64_t)n * h * wd * (c / 8) < (1ll << 40), "%s: tensor too large", who);
    return 0;
}

}  // namespace tsii

using namespace tsii;

extern "C" int64_t tsii_bf16_dw_stat_rows(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || c % 8 != 0 || kh != 3 || kw != 3 || sh != sw || (sh != 1 && sh != 2) || dh != dw || dh < 1 || (sh == 2 && dh != 1)) return 0;
    if (hdw_lean_ok(ho, wo, c, sh, dh)) return hdw_lean_rows(n, ho, wo, c, sh, dh);        // the lean strip kernel's layout (dw_lean_api.h)
    return hdw_rows(hdw_plan(n, 1, 1, c, ho, wo, sh, 0, dh, 0, 1));
}

extern "C" int tsii_bf16_dw_fwd(const uint16_t* x, const float* w, const float* bias, int n, int h, int wd, int c,
                                int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                float* stat_part, uint16_t* y, void* stream) {
    TSII_REQUIRE(x && w && y, "bf16_dw_fwd: null pointer");
    if (hdw_geom_ok("bf16_dw_fwd", n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(aligned16(x) && aligned16(y), "bf16_dw_fwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    InBN ib = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    if (in_scale != nullptr) {
        TSII_REQUIRE(in_shift != nullptr, "bf16_dw_fwd: in_scale / in_shift go together");
        TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "bf16_dw_fwd: activation %d has no load-time form", in_act);
    }
    if (hdw_lean_ok(ho, wo, c, sh, dh)) {      // stride 1, dilation 1 / 2 / 4: the LDS-slab strip kernel (round 6)
        const int rc = launch_hdw_lean(x, w, bias, n, h, wd, c, dh, ph, pw, ho, wo, 0, ib.sc, ib.sh, ib.neg, ib.hi, stat_part,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f, nullptr, y, st);
        if (rc <= 0) return rc;
        TSII_REQUIRE(stat_part == nullptr, "bf16_dw_fwd: this geometry is outside the strip kernel's limits but its partial rows were sized for it");
    }
    const HDwPlan g = hdw_plan(n, h, wd, c, ho, wo, sh, ph, dh, 0, 1);
    const dim3 grid((unsigned)hdw_blocks(g));
    const HDwBn nb = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f};
#define TSII_HDW(SS, BI, EP)                                                                                                              \
    do {                                                                                                                                   \
        if (SS == 1 && g.nx == 2) hipLaunchKernelGGL((hdw_conv_kernel<1, BI, EP, 2>), grid, dim3(256), 0, st, x, w, bias, g, ib, stat_part, nb, y); \
        else hipLaunchKernelGGL((hdw_conv_kernel<SS, BI, EP, 1>), grid, dim3(256), 0, st, x, w, bias, g, ib, stat_part, nb, y);           \
    } while (0)
    const bool bi = in_scale != nullptr, stt = stat_part != nullptr;
    if (sh == 1) { if (bi) { if (stt) TSII_HDW(1, true, 1); else TSII_HDW(1, true, 0); } else { if (stt) TSII_HDW(1, false, 1); else TSII_HDW(1, false, 0); } }
    else { if (bi) { if (stt) TSII_HDW(2, true, 1); else TSII_HDW(2, true, 0); } else { if (stt) TSII_HDW(2, false, 1); else TSII_HDW(2, false, 0); } }
#undef TSII_HDW
    return check_launch("bf16_dw_fwd");
}

extern "C" int64_t tsii_bf16_dw_bwd_stat_rows(int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 8 != 0 || kh != 3 || kw != 3 || sh != 1 || sw != 1 || dh != dw || dh < 1 || ph != pw) return 0;
    if (hdw_lean_ok(h, wd, c, 1, dh)) return hdw_lean_rows(n, h, wd, c, 1, dh);
    return hdw_rows(hdw_plan(n, 1, 1, c, h, wd, 1, 0, dh, 1, 2));
}

extern "C" int tsii_bf16_dw_bwd_dx(const uint16_t* dy, const float* w, int n, int h, int wd, int c,
                                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                   const uint16_t* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                   const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                   uint16_t* dx, float* bwd_part, void* stream) {
    TSII_REQUIRE(dy && w && dx, "bf16_dw_bwd_dx: null pointer");
    if (hdw_geom_ok("bf16_dw_bwd_dx", n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(aligned16(dy) && aligned16(dx), "bf16_dw_bwd_dx: tensors must be 16-byte aligned");
    TSII_REQUIRE((bn_y == nullptr) == (bwd_part == nullptr), "bf16_dw_bwd_dx: bn_y and bwd_part go together");
    hipStream_t st = (hipStream_t)stream;
    if (sh == 2) {
        TSII_REQUIRE(bn_y == nullptr, "bf16_dw_bwd_dx: the stride-2 form takes no BatchNorm-backward reductions (tsii_bf16_dw_bwd_stat_rows == 0)");
        TSII_REQUIRE(c <= HDW_S2_MAXC, "bf16_dw_bwd_dx: the stride-2 form is built for <= %d channels", HDW_S2_MAXC);
        const int64_t total = (int64_t)n * h * wd * (c / 8);
        hipLaunchKernelGGL(hdw_dx_s2_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, st, dy, w, n, h, wd, c, ho, wo, ph, dx);
        return check_launch("bf16_dw_bwd_dx (stride 2)");
    }
    // stride 1: the adjoint is the same stencil with flipped taps and padding 2 d - p over dy [n, ho, wo, c] -> dx [n, h, wd, c]
    if (hdw_lean_ok(h, wd, c, 1, dh) && 2 * dh - ph >= 0) {
        float neg = 1.f, hi = 0.f;
        if (bn_y != nullptr) {
            TSII_REQUIRE(bn_mean && bn_var && bn_gamma && bn_beta && aligned16(bn_y), "bf16_dw_bwd_dx: BatchNorm parameters missing");
            InBN tmp;
            TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "bf16_dw_bwd_dx: activation %d has no load-time form", bn_act);
            neg = tmp.neg; hi = tmp.hi;
        }
        const int rc = launch_hdw_lean(dy, w, nullptr, n, ho, wo, c, dh, 2 * dh - ph, 2 * dw - pw, h, wd, 1, nullptr, nullptr, 1.f, __builtin_huge_valf(), nullptr,
                                       bn_y, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, neg, hi, bwd_part, dx, st);
        if (rc <= 0) return rc;
        TSII_REQUIRE(bn_y == nullptr, "bf16_dw_bwd_dx: this geometry is outside the strip kernel's limits but its partial rows were sized for it");
    }
    const HDwPlan g = hdw_plan(n, ho, wo, c, h, wd, 1, 2 * dh - ph, dh, 1, bn_y != nullptr ? 2 : 1);
    const dim3 grid((unsigned)hdw_blocks(g));
    InBN ib = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    HDwBn kb = {nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f};
    if (bn_y != nullptr) {
        TSII_REQUIRE(bn_mean && bn_var && bn_gamma && bn_beta && aligned16(bn_y), "bf16_dw_bwd_dx: BatchNorm parameters missing");
        InBN tmp;
        TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "bf16_dw_bwd_dx: activation %d has no load-time form", bn_act);
        kb.y = bn_y; kb.mean = bn_mean; kb.var = bn_var; kb.gamma = bn_gamma; kb.beta = bn_beta; kb.eps = bn_eps; kb.neg = tmp.neg; kb.hi = tmp.hi;
        if (g.nx == 2) hipLaunchKernelGGL((hdw_conv_kernel<1, false, 2, 2>), grid, dim3(256), 0, st, dy, w, (const float*)nullptr, g, ib, bwd_part, kb, dx);
        else hipLaunchKernelGGL((hdw_conv_kernel<1, false, 2, 1>), grid, dim3(256), 0, st, dy, w, (const float*)nullptr, g, ib, bwd_part, kb, dx);
    } else {
        if (g.nx == 2) hipLaunchKernelGGL((hdw_conv_kernel<1, false, 0, 2>), grid, dim3(256), 0, st, dy, w, (const float*)nullptr, g, ib, (float*)nullptr, kb, dx);
        else hipLaunchKernelGGL((hdw_conv_kernel<1, false, 0, 1>), grid, dim3(256), 0, st, dy, w, (const float*)nullptr, g, ib, (float*)nullptr, kb, dx);
    }
    return check_launch("bf16_dw_bwd_dx");
}

extern "C" size_t tsii_bf16_dw_bwd_dw_ws_bytes(int n, int ho, int wo, int c, int kh, int kw, int sh, int sw, int dh, int dw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || c % 8 != 0 || kh != 3 || kw != 3 || (sh != 1 && sh != 2) || dh < 1) return 0;
    const HDwPlan g = hdw_plan(n, 1, 1, c, ho, wo, sh, 0, dh, 0, 0);
    const size_t rows = (size_t)hdw_rows(g);
    const size_t bias = (size_t)partial_rows((int64_t)n * ho * wo, c / 8) * c;
    const size_t fl = rows * 9 * c;
    return (fl > bias ? fl : bias) * sizeof(float) + 16;
}

extern "C" int tsii_bf16_dw_bwd_dw(const uint16_t* dy, const uint16_t* x, int n, int h, int wd, int c,
                                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                   const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                   float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && x && dwgt && ws, "bf16_dw_bwd_dw: null pointer");
    if (hdw_geom_ok("bf16_dw_bwd_dw", n, h, wd, c, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(aligned16(dy) && aligned16(x), "bf16_dw_bwd_dw: tensors must be 16-byte aligned");
    TSII_REQUIRE(ws_bytes >= tsii_bf16_dw_bwd_dw_ws_bytes(n, ho, wo, c, kh, kw, sh, sw, dh, dw), "bf16_dw_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    InBN ib = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    if (in_scale != nullptr) {
        TSII_REQUIRE(in_shift != nullptr, "bf16_dw_bwd_dw: in_scale / in_shift go together");
        TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "bf16_dw_bwd_dw: activation %d has no load-time form", in_act);
    }
    const HDwPlan g = hdw_plan(n, h, wd, c, ho, wo, sh, ph, dh, 0, 0);
    const dim3 grid((unsigned)hdw_blocks(g));
    float* wsf = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
    const bool bi = in_scale != nullptr;
    if (sh == 1 && g.nx == 2) { if (bi) hipLaunchKernelGGL((hdw_dw_kernel<1, true, 2>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); else hipLaunchKernelGGL((hdw_dw_kernel<1, false, 2>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); }
    else if (sh == 1) { if (bi) hipLaunchKernelGGL((hdw_dw_kernel<1, true, 1>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); else hipLaunchKernelGGL((hdw_dw_kernel<1, false, 1>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); }
    else { if (bi) hipLaunchKernelGGL((hdw_dw_kernel<2, true, 1>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); else hipLaunchKernelGGL((hdw_dw_kernel<2, false, 1>), grid, dim3(256), 0, st, x, dy, g, ib, wsf); }
    int rc = check_launch("bf16_dw_bwd_dw");
    if (rc) return rc;
    rc = launch_reduce_rows(wsf, (int)hdw_rows(g), (int64_t)c * 9, dwgt, st);
    if (rc) return rc;
    if (dbias != nullptr) return launch_bf16_colsum(dy, (int64_t)n * ho * wo, c, dbias, wsf, st);
    return 0;
}

extern "C" int tsii_bf16_avgpool(const uint16_t* x, int n, int h, int wd, int c, int k, uint16_t* y, void* stream) {
    TSII_REQUIRE(x && y && n > 0 && h > 0 && wd > 0 && c > 0 && c % 8 == 0, "bf16_avgpool: bad arguments (channels must be a multiple of 8)");
    TSII_REQUIRE(k == 3 || k == 5 || k == 9, "bf16_avgpool: k = 3, 5 or 9 (stride 1, padding (k-1)/2), got %d", k);
    TSII_REQUIRE(aligned16(x) && aligned16(y), "bf16_avgpool: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    HDwPlan g = hdw_plan(n, h, wd, c, h, wd, 2 /* one row phase, plain chunks */, 0, 1, 0, 0);
    const dim3 grid((unsigned)hdw_blocks(g));
    if (k == 3) hipLaunchKernelGGL(havgpool_kernel<3>, grid, dim3(256), 0, st, x, g, y);
    else if (k == 5) hipLaunchKernelGGL(havgpool_kernel<5>, grid, dim3(256), 0, st, x, g, y);
    else hipLaunchKernelGGL(havgpool_kernel<9>, grid, dim3(256), 0, st, x, g, y);
    return check_launch("bf16_avgpool");
}
