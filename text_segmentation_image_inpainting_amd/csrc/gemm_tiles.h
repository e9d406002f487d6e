// Tile loaders and the shared epilogue of the NT GEMM kernels (gemm.hip: fp32 MFMA; gemm_split.hip: split-bf16 MFMA).
#pragma once
#include "tsii_common.h"

#ifndef NT_EPI_NT_STORE
#define NT_EPI_NT_STORE 0     // A/B build knob: non-temporal stores of the output tile (shared LDS-staged epilogue)
#endif

namespace tsii {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int GEMM_BK = 32;
static constexpr int GEMM_LDS = GEMM_BK + 4;  // padded row stride (floats)

struct Epilogue {
    const float* denom;  // [M] divide
    const float* keep;   // [M] 0 -> output 0
    const float* bias;   // [N]
    RowScale cs;         // output scale per (row, col)
    int vec_store;       // C rows are 16-byte aligned (ldc % 4 == 0, base aligned)
    float* stats;        // [row blocks][4][N] BatchNorm partials of the stored values: (count, pivot, sum(y-pivot),
                         // sum((y-pivot)^2)), pivot = a value of the block itself (robust for near-constant channels); or NULL
    // K6c (dX feeding a BatchNorm backward; BNB instantiations only): the stored values are the gradient w.r.t.
    // a = act(gamma*xhat+beta) of the raw [M,N] tensor bn_y; the epilogue also leaves per row block (sum dz, sum dz*xhat)
    const float* bn_y;
    const float* bn_mean;
    const float* bn_var;
    const float* bn_gamma;
    const float* bn_beta;
    float bn_eps, bn_neg, bn_hi;
    float* bn_part;      // [row blocks][2][N]
    // Forward with an up-sampled addend (round 4): rows are the pixels of [.., up_h, up_w] images and the accumulator of row
    // (n, y, x) gets up_add[(n, y / 2, x / 2)][col] (a [M / 4, N] matrix) added BEFORE the count division / bias / hole zeroing:
    // the low-resolution half of a 1x1 convolution over cat(nearest-x2(low), skip), computed at low resolution.  NULL: none.
    const float* up_add;
    unsigned up_w, up_magic, up_shift;   // row / up_w = umulhi(row, up_magic) >> up_shift for row < 2^31 (make_up_div)
};

// exact unsigned division by a constant for numerators below 2^31: q = umulhi(n, magic) >> shift
static inline void make_up_div(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                               // s = ceil(log2 d), d >= 2
    *magic = (unsigned)(((1ull << (31 + s)) + d - 1) / d);     // < 2^32; error term n * e < 2^(31+s) for n < 2^31
    *shift = s - 1;
}
// row of the [M/4, N] addend for GEMM row `row` (pixel (n, y, x) of images up_w wide, even height)
__device__ __forceinline__ unsigned up_low_row(unsigned row, unsigned w, unsigned magic, unsigned shift) {
    const unsigned q = __umulhi(row, magic) >> shift;          // n * H + y
    const unsigned x = row - q * w;
    return (q >> 1) * (w >> 1) + (x >> 1);
}


// ---- tile loaders ----------------------------------------------------------------------
// NT: tile of ROWS x 32 floats from a row-major matrix (K contiguous); 8 float4 per row.  A thread
// touches the same ROWS/32 rows in every K-tile, so its row scales (s0 for k < split, s1 after)
// are loaded once before the K loop.
template <int ROWS, bool VEC>
__device__ __forceinline__ void nt_load(const float* __restrict__ P, int64_t ld, int64_t row0, int64_t nrows,
                                        int k0, int K, float4 (&regs)[ROWS / 32]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        const int r = f >> 3, c4 = f & 7;
        const int64_t row = row0 + r;
        const int k = k0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nrows) {
            const float* p = P + row * ld + k;
            if (VEC) {
                if (k < K) v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k + 0 < K) v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
        regs[i] = v;   // NOT scaled here: a use of the loaded value would force vmcnt(0) before the MFMAs
    }
}

template <int ROWS>
__device__ __forceinline__ void nt_row_scales(const RowScale& rs, int64_t row0, int64_t nrows,
                                              float (&s0)[ROWS / 32], float (&s1)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int64_t row = row0 + ((threadIdx.x + 256 * i) >> 3);
        s0[i] = 1.f; s1[i] = 1.f;
        if (rs.r0 != nullptr && row < nrows) {
            s0[i] = rs.r0[row];
            s1[i] = rs.r1 != nullptr ? rs.r1[row] : 1.f;
        }
    }
}

// registers -> LDS after the MFMA phase; the x*mask row scale (s0 for k < split, s1 after) rides here
template <int ROWS, bool SCALED>
__device__ __forceinline__ void nt_store(float* __restrict__ S, const float4 (&regs)[ROWS / 32], int k0, int split,
                                         const float (&s0)[ROWS / 32], const float (&s1)[ROWS / 32]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        const int r = f >> 3, c4 = f & 7;
        float4 v = regs[i];
        if (SCALED) {
            const int k = k0 + c4 * 4;
            v.x *= (k + 0 < split) ? s0[i] : s1[i];
            v.y *= (k + 1 < split) ? s0[i] : s1[i];
            v.z *= (k + 2 < split) ? s0[i] : s1[i];
            v.w *= (k + 3 < split) ? s0[i] : s1[i];
        }
        *reinterpret_cast<float4*>(S + r * GEMM_LDS + c4 * 4) = v;
    }
}

// nt_store with the producer's BatchNorm + activation applied first (psc/psh: this thread's 4 channels of the tile)
template <int ROWS>
__device__ __forceinline__ void nt_store_bn(float* __restrict__ S, const float4 (&regs)[ROWS / 32], int k0, int split,
                                            const float (&s0)[ROWS / 32], const float (&s1)[ROWS / 32], const float4 psc,
                                            const float4 psh, float neg, float hi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        const int r = f >> 3, c4 = f & 7;
        const int k = k0 + c4 * 4;
        float4 v = regs[i];
        v.x = bn_act_load(v.x, psc.x, psh.x, neg, hi) * ((k + 0 < split) ? s0[i] : s1[i]);
        v.y = bn_act_load(v.y, psc.y, psh.y, neg, hi) * ((k + 1 < split) ? s0[i] : s1[i]);
        v.z = bn_act_load(v.z, psc.z, psh.z, neg, hi) * ((k + 2 < split) ? s0[i] : s1[i]);
        v.w = bn_act_load(v.w, psc.w, psh.w, neg, hi) * ((k + 3 < split) ? s0[i] : s1[i]);
        *reinterpret_cast<float4*>(S + r * GEMM_LDS + c4 * 4) = v;
    }
}

// ---- implicit-GEMM gather (dense k x k convolutions on the MFMA path) ------------------------------
// The A operand of the NT kernel (and the B operand of the TN kernel) can be an im2col VIEW of an NHWC tensor:
// row m = a pixel of the row grid [n, rh, rw], column k = (tap t, source channel ci), t = ky*kw + kx.
//   AMODE 1 (forward / dW):  source pixel = (ry*sh - ph + ky*dh, rx*sw - pw + kx*dw)
//   AMODE 2 (dX):            source pixel = ((ry + ph - ky*dh)/sh, (rx + pw - kx*dw)/sw) when divisible
// Out-of-range taps contribute zeros (zero padding).  p0/p1: per-source-pixel planes multiplied in at the LDS
// store (x*mask with a channel split for forward/dW, 1/count for dX).
struct ConvGather {
    int h, w, c;     // source tensor [n, h, w, c]
    int rh, rw;      // row grid
    int kw;          // kernel width
    int sh, sw, ph, pw, dh, dw;
    const float* p0;
    const float* p1;
    int split;       // channels < split use p0, the rest p1 (p1 == nullptr -> 1.0)
    const float* pfull;  // general per-channel mask with the source tensor's shape (element-wise gather only)
    // dX of a strided conv, one launch per stride phase (AMODE 2): row (n, ry, rx) of the phase grid is the output pixel
    // (n, ry*o_sy + o_y0, rx*o_sx + o_x0) of the [.., o_h, o_w] tensor; o_sy == 0: rows are output pixels as they come
    int o_h, o_w, o_sy, o_sx, o_y0, o_x0;
};

template <int AMODE>
__device__ __forceinline__ bool conv_src(const ConvGather& cg, int ry, int rx, int ky, int kx, int& sy, int& sx) {
    if (AMODE == 1) {
        sy = ry * cg.sh - cg.ph + ky * cg.dh;
        sx = rx * cg.sw - cg.pw + kx * cg.dw;
        return sy >= 0 && sy < cg.h && sx >= 0 && sx < cg.w;
    }
    const int ty = ry + cg.ph - ky * cg.dh, tx = rx + cg.pw - kx * cg.dw;
    if (ty < 0 || tx < 0 || (ty % cg.sh) != 0 || (tx % cg.sw) != 0) return false;
    sy = ty / cg.sh; sx = tx / cg.sw;
    return sy < cg.h && sx < cg.w;
}

// element-wise form of the gather for channel counts that are not a multiple of 4 (3-channel stems) and for
// per-channel masks: every k decodes its own (tap, channel); the mask factor is applied at once (the K loop of
// such layers is a handful of tiles, so the early use of the loaded value does not matter).
__device__ __forceinline__ float conv_gather_elem(const float* __restrict__ src, const ConvGather& cg, int n, int ry, int rx, int k) {
    const int t = k / cg.c, ci = k - t * cg.c;
    const int ky = t / cg.kw, kx = t - ky * cg.kw;
    int sy, sx;
    if (!conv_src<1>(cg, ry, rx, ky, kx, sy, sx)) return 0.f;
    const int64_t spix = ((int64_t)n * cg.h + sy) * cg.w + sx;
    float v = src[spix * cg.c + ci];
    if (cg.pfull != nullptr) v *= cg.pfull[spix * cg.c + ci];
    else if (cg.p0 != nullptr) v *= (ci < cg.split) ? cg.p0[spix] : (cg.p1 != nullptr ? cg.p1[spix] : 1.f);
    return v;
}

// rows of this thread (fixed across K tiles): image index (or -1), y, x on the row grid
template <int ROWS>
__device__ __forceinline__ void conv_rows(const ConvGather& cg, int64_t row0, int64_t nrows, int (&rn)[ROWS / 32],
                                          int (&ry)[ROWS / 32], int (&rx)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int64_t row = row0 + ((threadIdx.x + 256 * i) >> 3);
        rn[i] = -1; ry[i] = 0; rx[i] = 0;
        if (row < nrows) {
            rx[i] = (int)(row % cg.rw);
            ry[i] = (int)((row / cg.rw) % cg.rh);
            rn[i] = (int)(row / ((int64_t)cg.rw * cg.rh));
        }
    }
}

template <int ROWS, int AMODE>
__device__ __forceinline__ void conv_load(const float* __restrict__ src, const ConvGather& cg, const int (&rn)[ROWS / 32],
                                          const int (&ry)[ROWS / 32], const int (&rx)[ROWS / 32], int k0, int K,
                                          float4 (&regs)[ROWS / 32], float (&f0)[ROWS / 32], float (&f1)[ROWS / 32]) {
    const int k = k0 + (threadIdx.x & 7) * 4;          // same column group for all of this thread's rows
    if constexpr (AMODE == 3) {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rn[i] >= 0) {
                if (k + 0 < K) v.x = conv_gather_elem(src, cg, rn[i], ry[i], rx[i], k + 0);
                if (k + 1 < K) v.y = conv_gather_elem(src, cg, rn[i], ry[i], rx[i], k + 1);
                if (k + 2 < K) v.z = conv_gather_elem(src, cg, rn[i], ry[i], rx[i], k + 2);
                if (k + 3 < K) v.w = conv_gather_elem(src, cg, rn[i], ry[i], rx[i], k + 3);
            }
            regs[i] = v; f0[i] = 1.f; f1[i] = 1.f;
        }
        return;
    }
    const int t = k / cg.c, ci = k - t * cg.c;
    const int ky = t / cg.kw, kx = t - ky * cg.kw;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float a0 = 1.f, a1 = 1.f;
        int sy, sx;
        if (k < K && rn[i] >= 0 && conv_src<(AMODE == 2 ? 2 : 1)>(cg, ry[i], rx[i], ky, kx, sy, sx)) {
            const int64_t spix = ((int64_t)rn[i] * cg.h + sy) * cg.w + sx;
            v = *reinterpret_cast<const float4*>(src + spix * cg.c + ci);
            if (cg.p0 != nullptr) { a0 = cg.p0[spix]; a1 = cg.p1 != nullptr ? cg.p1[spix] : 1.f; }
        }
        regs[i] = v; f0[i] = a0; f1[i] = a1;
    }
}

template <int ROWS>
__device__ __forceinline__ void conv_store(float* __restrict__ S, const float4 (&regs)[ROWS / 32], const ConvGather& cg, int k0,
                                           const float (&f0)[ROWS / 32], const float (&f1)[ROWS / 32]) {
    const int tid = threadIdx.x;
    const int c4 = tid & 7;
    const int ci = (k0 + c4 * 4) % cg.c;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int r = (tid + 256 * i) >> 3;
        float4 v = regs[i];
        v.x *= (ci + 0 < cg.split) ? f0[i] : f1[i];
        v.y *= (ci + 1 < cg.split) ? f0[i] : f1[i];
        v.z *= (ci + 2 < cg.split) ? f0[i] : f1[i];
        v.w *= (ci + 3 < cg.split) ? f0[i] : f1[i];
        *reinterpret_cast<float4*>(S + r * GEMM_LDS + c4 * 4) = v;
    }
}

// ---- split-bf16 kernels (gemm_split.hip) ----------------------------------------------------------------
int gemm_products();     // 0: f32-input MFMA | 3 / 6: split-bf16 partial products (tsii_set_gemm_products)
bool nt_split_ok(const float* A, int64_t lda, const float* B, int64_t ldb, int K);
size_t nt_split_ws_bytes(int n, int k);
int launch_nt_split(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, bool b_transposed, float* C, int64_t ldc,
                    int64_t M, int N, int K, Epilogue ep, InBN ib, void* wsplit, hipStream_t stream);

// producer / consumer form of the NT split kernel (gemm_pc.hip): plain 1x1 layers in the 6-product mode
bool nt_pc_ok(const float* A, int64_t lda, const RowScale& as, int64_t M, int N, int K, const Epilogue& ep, const InBN& ib);
size_t nt_pc_ws_bytes(int n, int k);
int launch_nt_pc(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, bool b_transposed, float* C, int64_t ldc,
                 int64_t M, int N, int K, Epilogue ep, InBN ib, void* wsplit, hipStream_t stream);

// dense convolutions on the split kernels: A gathered by cg (amode 1 forward, 2 dX; cg.c % 4 == 0, K % 8 == 0), B = fp32 [N,K]
bool nt_split_conv_ok(int amode, const float* A, const float* B, int64_t ldb, int K, const ConvGather& cg);
int launch_nt_split_conv(int amode, const float* A, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                         Epilogue ep, const ConvGather& cg, hipStream_t stream);

bool tn_split_ok(const float* A, int64_t lda, const float* B, int64_t ldb, int Pn, int Q);
// dW of a dense convolution on the split TN kernel: B gathered by cg (cg.c % 4 == 0); big: 128x128 tiles, else 64x64 (plain-bf16 mode only)
bool tn_split_conv_ok(const float* A, int64_t lda, const float* B, int64_t M, int Pn, int Q, const ConvGather& cg);
int launch_tn_split_conv(const float* A, int64_t lda, const float* sa, const float* B, const ConvGather& cg, float* Cws, int64_t M, int Pn, int Q,
                         int64_t chunk, int splits, bool big, hipStream_t stream);
int launch_tn_split(const float* A, int64_t lda, const float* sa, const float* B, int64_t ldb, RowScale sb, float* Cws,
                    int64_t M, int Pn, int Q, int64_t chunk, int splits, int tile, InBN ib, hipStream_t stream);

// ---- epilogue shared by the NT kernels -----------------------------------------------------------------
// acc[t][u] = 32x32 MFMA accumulators of the wave (D[row=(r&3)+8*(r>>2)+4*hi][col=lane&31], the same map for the
// f32 32x32x2 and the bf16 32x32x16 instruction); smem = the block's operand LDS, SMEM_FLOATS floats, free to reuse.
template <int WM, int WN, int TM, int TN, int AMODE, bool BNB, int SMEM_FLOATS>
__device__ __forceinline__ void nt_epilogue(float* __restrict__ smem, f32x16 (&acc)[TM][TN], float* __restrict__ C, int64_t ldc,
                                            int64_t M, int N, const Epilogue& ep, const ConvGather& cg, int64_t m0, int n0,
                                            unsigned bid, unsigned ntn) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;
    // epilogue: accumulators (D[row=(r&3)+8*(r>>2)+4*hi][col=lane&31]) are staged through LDS one
    // 32-row band per wave-row at a time, so rows leave as 16-byte stores (the dword-per-lane form is
    // store-issue bound for the short-K layers); count division / bias / hole zeroing ride along.
    // A thread owns ONE 4-column group (256 % F4_PER_ROW == 0) and F4_PER_THREAD rows per band; all the
    // per-row side loads of a band are issued together before the first use -- interleaved load/use made
    // the epilogue a chain of ~8 dependent L2 round trips per float4 and capped short-K layers at ~60 TF/s.
    constexpr int CS = BN + 4;                       // padded LDS row stride (floats), keeps float4 alignment
    constexpr int BAND_ROWS = WM * 32;
    constexpr int F4_PER_ROW = BN / 4;
    constexpr int F4_PER_THREAD = BAND_ROWS * F4_PER_ROW / 256;
    constexpr int ROW_STEP = 256 / F4_PER_ROW;
    static_assert(BAND_ROWS * CS <= SMEM_FLOATS, "epilogue band must fit the operand LDS");
    static_assert(256 % F4_PER_ROW == 0, "fixed column group per thread");
    float* Cs = smem;
    const int c4 = tid % F4_PER_ROW, rr0 = tid / F4_PER_ROW;
    const int col = n0 + c4 * 4;
    const bool col_ok = col < N;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < N) bv[e] = ep.bias[col + e];
    }
    const bool has_cs = ep.cs.r0 != nullptr;
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};   // BatchNorm partial sums about the block pivot
    float pvt[4] = {0.f, 0.f, 0.f, 0.f};
    float bmu[4] = {0.f, 0.f, 0.f, 0.f}, bis[4] = {0.f, 0.f, 0.f, 0.f}, bga[4] = {0.f, 0.f, 0.f, 0.f}, bbe[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BNB) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < N) {
                bmu[e] = ep.bn_mean[col + e]; bis[e] = 1.0f / sqrtf(ep.bn_var[col + e] + ep.bn_eps);
                bga[e] = ep.bn_gamma[col + e]; bbe[e] = ep.bn_beta[col + e];
            }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        // side loads of this band (clamped rows: no divergent branches, nothing is used until the tile is staged)
        float kpv[F4_PER_THREAD], dnv[F4_PER_THREAD], c0v[F4_PER_THREAD], c1v[F4_PER_THREAD];
        int64_t rowv[F4_PER_THREAD];
#pragma unroll
        for (int i = 0; i < F4_PER_THREAD; ++i) {
            const int rr = rr0 + ROW_STEP * i;
            rowv[i] = m0 + ((rr >> 5) * TM + t) * 32 + (rr & 31);
            kpv[i] = 1.f; dnv[i] = 1.f; c0v[i] = 1.f; c1v[i] = 1.f;
            if constexpr (AMODE == 2) {
                if (cg.o_sy != 0) {          // stride-phase launch: GEMM row -> pixel of the strided output (rows >= M stay >= M)
                    const int64_t row = rowv[i];
                    if (row < M) {
                        const int rx = (int)(row % cg.rw), ry = (int)((row / cg.rw) % cg.rh);
                        const int64_t n = row / ((int64_t)cg.rw * cg.rh);
                        rowv[i] = (n * cg.o_h + ry * cg.o_sy + cg.o_y0) * (int64_t)cg.o_w + rx * cg.o_sx + cg.o_x0;
                    } else {
                        rowv[i] = INT64_MAX;
                    }
                }
            }
        }
        const int64_t Mout = (AMODE == 2 && cg.o_sy != 0) ? (int64_t)INT64_MAX : M;   // remapped rows are already range-checked
        if (ep.keep != nullptr) {
#pragma unroll
            for (int i = 0; i < F4_PER_THREAD; ++i) kpv[i] = ep.keep[rowv[i] < Mout ? rowv[i] : 0];
        }
        if (ep.denom != nullptr) {
#pragma unroll
            for (int i = 0; i < F4_PER_THREAD; ++i) dnv[i] = ep.denom[rowv[i] < Mout ? rowv[i] : 0];
        }
        if (has_cs) {
#pragma unroll
            for (int i = 0; i < F4_PER_THREAD; ++i) c0v[i] = ep.cs.r0[rowv[i] < Mout ? rowv[i] : 0];
            if (ep.cs.r1 != nullptr) {
#pragma unroll
                for (int i = 0; i < F4_PER_THREAD; ++i) c1v[i] = ep.cs.r1[rowv[i] < Mout ? rowv[i] : 0];
            }
        }
        float4 zq[F4_PER_THREAD];                              // up-sampled addend (N % 4 == 0, checked by the entry point)
        if (ep.up_add != nullptr) {
#pragma unroll
            for (int i = 0; i < F4_PER_THREAD; ++i) {
                zq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowv[i] < Mout && col_ok)
                    zq[i] = *reinterpret_cast<const float4*>(ep.up_add + (int64_t)up_low_row((unsigned)rowv[i], ep.up_w, ep.up_magic, ep.up_shift) * N + col);
            }
        }
        float4 yq[BNB ? F4_PER_THREAD : 1];
        if constexpr (BNB) {     // the raw BatchNorm input at the positions this thread stores (N % 4 == 0, 16-byte rows)
#pragma unroll
            for (int i = 0; i < F4_PER_THREAD; ++i) {
                yq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowv[i] < Mout && col_ok) yq[i] = *reinterpret_cast<const float4*>(ep.bn_y + rowv[i] * (int64_t)N + col);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CS + (wn * TN + u) * 32 + li] = acc[t][u][r];
        __syncthreads();
        if (t == 0 && ep.stats != nullptr) {
            // block pivot: (roughly) the value of the block's middle row (row 64; image-border pixels, where the first
            // row of a block often sits, are the outliers of a channel) -- any number typical of the column does, it
            // only has to be the same for every thread of the column group
            const bool mid = m0 + 64 < M;
            const float4 q0 = *reinterpret_cast<const float4*>(Cs + (mid ? (2 / TM) * 32 * CS : 0) + c4 * 4);
            const float rd0 = ep.denom != nullptr ? 1.0f / ep.denom[mid ? m0 + 64 : m0] : 1.0f;
            pvt[0] = fmaf(q0.x, rd0, bv[0]); pvt[1] = fmaf(q0.y, rd0, bv[1]);
            pvt[2] = fmaf(q0.z, rd0, bv[2]); pvt[3] = fmaf(q0.w, rd0, bv[3]);
        }
#pragma unroll
        for (int i = 0; i < F4_PER_THREAD; ++i) {
            const int rr = rr0 + ROW_STEP * i;
            const int64_t row = rowv[i];
            if (row >= Mout || !col_ok) continue;
            const float4 q = *reinterpret_cast<const float4*>(Cs + rr * CS + c4 * 4);
            float v[4] = {q.x, q.y, q.z, q.w};
            if (ep.up_add != nullptr) { v[0] += zq[i].x; v[1] += zq[i].y; v[2] += zq[i].z; v[3] += zq[i].w; }
            if (ep.denom != nullptr) {
                const float rd = 1.0f / dnv[i];      // one IEEE division per row, then multiplies (<= 1 ulp apart)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= rd;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += bv[e];
                if (kpv[i] == 0.f) v[e] = 0.f;
                if (has_cs) v[e] *= (col + e < ep.cs.split) ? c0v[i] : c1v[i];
            }
            if (ep.stats != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[e] - pvt[e];
                    st1[e] += d;
                    st2[e] = fmaf(d, d, st2[e]);
                }
            }
            if constexpr (BNB) {
                const float ye[4] = {yq[i].x, yq[i].y, yq[i].z, yq[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (ye[e] - bmu[e]) * bis[e];
                    const float z = fmaf(xh, bga[e], bbe[e]);
                    const float dz = v[e] * ((z > 0.f && z < ep.bn_hi) ? 1.f : (z > 0.f ? 0.f : ep.bn_neg));
                    st1[e] += dz;
                    st2[e] = fmaf(dz, xh, st2[e]);
                }
            }
            float* cp = C + row * ldc + col;
            if (ep.vec_store && col + 3 < N) {
                if (NT_EPI_NT_STORE) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(cp));
                else *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) cp[e] = v[e];
            }
        }
    }
    if constexpr (BNB) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Cs[(rr0 * BN + c4 * 4 + e) * 2 + 0] = st1[e];
            Cs[(rr0 * BN + c4 * 4 + e) * 2 + 1] = st2[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
            for (int j = 0; j < ROW_STEP; ++j) { a1 += Cs[(j * BN + tid) * 2 + 0]; a2 += Cs[(j * BN + tid) * 2 + 1]; }
            float* sp = ep.bn_part + (int64_t)(bid / ntn) * 2 * N;
            sp[n0 + tid] = a1;
            sp[N + n0 + tid] = a2;
        }
        return;
    }
    if (ep.stats != nullptr) {
        // ROW_STEP threads share a column group: combine through LDS, one partial row per row block
        static_assert(ROW_STEP * BN * 2 + BN <= SMEM_FLOATS, "stat reduction must fit the operand LDS");
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Cs[(rr0 * BN + c4 * 4 + e) * 2 + 0] = st1[e];
            Cs[(rr0 * BN + c4 * 4 + e) * 2 + 1] = st2[e];
        }
        if (rr0 == 0) {   // pivots: one writer per column group, region after the sums
#pragma unroll
            for (int e = 0; e < 4; ++e) Cs[ROW_STEP * BN * 2 + c4 * 4 + e] = pvt[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
            for (int j = 0; j < ROW_STEP; ++j) { a1 += Cs[(j * BN + tid) * 2 + 0]; a2 += Cs[(j * BN + tid) * 2 + 1]; }
            float* sp = ep.stats + (int64_t)(bid / ntn) * 4 * N;
            const int64_t left = M - m0;
            sp[n0 + tid] = (float)(left < BM ? left : BM);
            sp[N + n0 + tid] = Cs[ROW_STEP * BN * 2 + tid];
            sp[2 * N + n0 + tid] = a1;
            sp[3 * N + n0 + tid] = a2;
        }
    }
}

}  // namespace tsii
