// K3s: the point-wise GEMMs on the bf16 matrix cores with fp32-class accuracy ("split-bf16").
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA (2.5 PF/s vs 157 TF/s), and ImageFill's
// 1x1 convolutions sit right at the fp32 ridge (arithmetic intensity 25..130 F/B against 157 TF/s / 8 TB/s = 20 F/B):
// on the f32 MFMA every one of them is matrix-core bound, on the bf16 MFMA the high-resolution ones become HBM bound.
// fp32 parity (1e-3, and the BatchNorm chains amplify rounding noise ~1e4x) rules out plain bf16 inputs,
// so each fp32 operand is split EXACTLY into bf16 pieces while it is staged into LDS,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (RNE, 3 x 8 significand bits)
// and the product is assembled from the partial products whose weight is >= 2^-16 relative:
//     PRODUCTS = 6:  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)      dropped terms <= 2^-23 |ab|  (fp32 class)
//     PRODUCTS = 3:  a0b0 + (a0b1 + a1b0)                  [2 planes]  dropped terms <= 2^-15 |ab|  (opt-in, inference)
// accumulated in fp32 by the MFMA (small terms first).  6 bf16 MFMAs of K = 16 replace 8 f32 MFMAs of K = 2:
// 192 instead of 512 matrix-pipe cycles per 32x32x16 block.  HBM traffic is unchanged (fp32 in, fp32 out); the split
// costs ~5.5 VALU ops per element in the staging pass, which runs on the otherwise idle vector pipe.
//
// Structure (same as gemm.hip so the loaders' row scales / BatchNorm-on-load and the whole epilogue are shared):
// 256 threads = 4 waves, 128x128 (128x64, 128x32) tile, BK = 32, global -> registers -> split -> LDS with the next
// tile's loads in flight across the MFMA phase.  LDS image per operand: [plane][row][32 bf16] (64-byte rows, no padding)
// with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3, which makes both the ds_write_b128 of the staging pass
// (8-lane groups: 2 rows x 4 chunks) and the ds_read_b128 of the fragments (lane l: row l & 31, k-chunk 2s + (l >> 5);
// 16-lane groups {0-3,12-15,20-27}...) bank-conflict free.  48 KB per block at 3 planes -> 3 blocks per CU.
#include <stdlib.h>

#include "gemm_tiles.h"
#include "split_bf16.h"

namespace tsii {

// chunks of a ROWS x 32 tile handled per thread: f = tid + 256*i -> row f >> 2, chunk f & 3 (= tid & 3 for every i)
template <int ROWS>
struct SplitChunks { static constexpr int value = (ROWS * 4 + 255) / 256; };

// raw loads of this thread's chunks: 8 consecutive k of a row-major [rows, K] fp32 matrix as 2 float4.  BRANCH-FREE:
// out-of-range rows / k are clamped to valid addresses and zeroed by the store pass (a branch around each load made
// hipcc wait vmcnt(0) after every single load -- the loads must issue back to back and stay in flight over the MFMAs).
template <int ROWS>
__device__ __forceinline__ void split_load(const float* __restrict__ Pm, int64_t ld, int64_t row0, int64_t nrows, int k0, int K,
                                           float4 (&regs)[SplitChunks<ROWS>::value][2]) {
    // wave-uniform 64-bit base (SGPRs) + 32-bit per-thread BYTE offsets (saddr + voffset addressing)
    const int tid = threadIdx.x;
    const char* __restrict__ base = reinterpret_cast<const char*>(Pm + row0 * ld);
    const int last = (int)((nrows - row0 < ROWS) ? (nrows - row0) : ROWS) - 1;      // >= 0: the block has at least one row
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        int r = f >> 2;
        r = r < last ? r : last;
        int k = k0 + (f & 3) * 8;
        k = k < K - 8 ? k : K - 8;                                                 // K % 8 == 0, K >= 8
        const float* p = reinterpret_cast<const float*>(base + (unsigned)((r * (int)ld + k) * 4));
        regs[i][0] = *reinterpret_cast<const float4*>(p);
        regs[i][1] = *reinterpret_cast<const float4*>(p + 4);
    }
}

template <int ROWS>
__device__ __forceinline__ void split_row_scales(const RowScale& rs, int64_t row0, int64_t nrows,
                                                 float (&s0)[SplitChunks<ROWS>::value], float (&s1)[SplitChunks<ROWS>::value]) {
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int64_t row = row0 + ((threadIdx.x + 256 * i) >> 2);
        s0[i] = 1.f; s1[i] = 1.f;
        if (rs.r0 != nullptr && row < nrows) {
            s0[i] = rs.r0[row];
            s1[i] = rs.r1 != nullptr ? rs.r1[row] : 1.f;
        }
    }
}

// registers -> (BatchNorm + activation of the producer) -> x*mask row scale -> bf16 planes -> LDS
template <int ROWS, int P, bool SCALED, bool BNIN>
__device__ __forceinline__ void split_store(unsigned char* __restrict__ S, const float4 (&regs)[SplitChunks<ROWS>::value][2], int k0, int split,
                                            const float (&s0)[SplitChunks<ROWS>::value], const float (&s1)[SplitChunks<ROWS>::value],
                                            const float4 (&psc)[2], const float4 (&psh)[2], float neg, float hi, int nvalid, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const int r = f >> 2, c = f & 3;
        float v[8] = {regs[i][0].x, regs[i][0].y, regs[i][0].z, regs[i][0].w, regs[i][1].x, regs[i][1].y, regs[i][1].z, regs[i][1].w};
        const bool valid = r < nvalid && k0 + c * 8 < K;      // clamped loads (split_load): rows / k outside the matrix are zeros
        if constexpr (BNIN) {
            const float sc[8] = {psc[0].x, psc[0].y, psc[0].z, psc[0].w, psc[1].x, psc[1].y, psc[1].z, psc[1].w};
            const float sh[8] = {psh[0].x, psh[0].y, psh[0].z, psh[0].w, psh[1].x, psh[1].y, psh[1].z, psh[1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bn_act_load(v[e], sc[e], sh[e], neg, hi);
        }
        if constexpr (SCALED) {
            const int k = k0 + c * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= (k + e < split) ? s0[i] : s1[i];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = valid ? v[e] : 0.f;
        u32x4 pl[P];
        split8<P>(v, pl);
        const int off = split_off(r, c);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (ROWS * 64) + off) = pl[p];
    }
}

// ---- implicit-GEMM gather for the A operand (dense k x k convolutions, gemm_tiles.h ConvGather) ---------------------
// Row m = a pixel of the row grid, column k = (tap, channel); c % 4 == 0: a chunk of 8 k is two float4 gathers (one tap when
// c % 8 == 0).  This thread's rows are fixed for the tile (decoded once), its chunk column
// advances with the K loop.  Branch-free like split_load: taps outside the image read pixel 0 and are zeroed by the store
// (bit i of `okm`); the per-source-pixel planes p0 / p1 (x*mask, 1/count) arrive next to the data.
template <int ROWS>
__device__ __forceinline__ void split_conv_rows(const ConvGather& cg, int64_t row0, int64_t nrows, int (&rn)[SplitChunks<ROWS>::value],
                                                int (&ry)[SplitChunks<ROWS>::value], int (&rx)[SplitChunks<ROWS>::value]) {
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int64_t row = row0 + ((threadIdx.x + 256 * i) >> 2);
        rn[i] = -1; ry[i] = 0; rx[i] = 0;
        if (row < nrows) {
            rx[i] = (int)(row % cg.rw);
            ry[i] = (int)((row / cg.rw) % cg.rh);
            rn[i] = (int)(row / ((int64_t)cg.rw * cg.rh));
        }
    }
}

template <int ROWS, int AMODE>
__device__ __forceinline__ void split_conv_load(const float* __restrict__ src, const ConvGather& cg, const int (&rn)[SplitChunks<ROWS>::value],
                                                const int (&ry)[SplitChunks<ROWS>::value], const int (&rx)[SplitChunks<ROWS>::value],
                                                int k0, int K, float4 (&regs)[SplitChunks<ROWS>::value][2],
                                                float (&f0)[SplitChunks<ROWS>::value][2], float (&f1)[SplitChunks<ROWS>::value][2],
                                                unsigned& okm, int (&ci_out)[2]) {
    // the chunk's two float4 halves are gathered separately: with c % 8 == 0 they are neighbours inside one tap (decoded
    // once), with c % 4 == 0 only (12-channel space-to-depth stems) the second half may belong to the next tap
    int k = k0 + (threadIdx.x & 3) * 8;
    const bool kok = k < K;
    k = kok ? k : K - 8;                                                // K % 8 == 0
    const bool c8 = (cg.c & 7) == 0;                                    // wave-uniform
    okm = 0u;
    int t = k / cg.c, ci = k - t * cg.c;
    int ky = t / cg.kw, kx = t - ky * cg.kw;
    constexpr bool ONE_TAP = (AMODE == 2);          // dX gathers over cout: the launcher admits c % 8 == 0 only (one decode)
    int64_t spix0[SplitChunks<ROWS>::value];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) {
            ci += 4;
            if (!ONE_TAP && !c8 && ci >= cg.c) { ci -= cg.c; ++kx; if (kx == cg.kw) { kx = 0; ++ky; } }
        }
        ci_out[h] = ci;
#pragma unroll
        for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
            if ((ONE_TAP || c8) && h == 1) {          // same source pixel as the first half (wave-uniform test)
                regs[i][1] = *reinterpret_cast<const float4*>(src + spix0[i] * cg.c + ci);
                f0[i][1] = f0[i][0]; f1[i][1] = f1[i][0];
                okm |= ((okm >> (2 * i)) & 1u) << (2 * i + 1);
                continue;
            }
            int sy = 0, sx = 0;
            const bool ok = kok && rn[i] >= 0 && conv_src<AMODE>(cg, ry[i], rx[i], ky, kx, sy, sx);
            const int64_t spix = ok ? ((int64_t)rn[i] * cg.h + sy) * cg.w + sx : 0;
            if (h == 0) spix0[i] = spix;
            regs[i][h] = *reinterpret_cast<const float4*>(src + spix * cg.c + ci);
            float a0 = 1.f, a1 = 1.f;
            if (cg.p0 != nullptr) {                                      // wave-uniform
                a0 = cg.p0[spix];
                a1 = cg.p1 != nullptr ? cg.p1[spix] : 1.f;
            }
            f0[i][h] = a0; f1[i][h] = a1;
            okm |= ok ? (1u << (2 * i + h)) : 0u;
        }
    }
}

template <int ROWS, int P>
__device__ __forceinline__ void split_conv_store(unsigned char* __restrict__ S, const float4 (&regs)[SplitChunks<ROWS>::value][2], const int (&ci)[2],
                                                 int split, const float (&f0)[SplitChunks<ROWS>::value][2],
                                                 const float (&f1)[SplitChunks<ROWS>::value][2], unsigned okm) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const int r = f >> 2, c = f & 3;
        float v[8] = {regs[i][0].x, regs[i][0].y, regs[i][0].z, regs[i][0].w, regs[i][1].x, regs[i][1].y, regs[i][1].z, regs[i][1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int h = e >> 2;
            v[e] *= (ci[h] + (e & 3) < split) ? f0[i][h] : f1[i][h];
            v[e] = ((okm >> (2 * i + h)) & 1u) ? v[e] : 0.f;
        }
        u32x4 pl[P];
        split8<P>(v, pl);
        const int off = split_off(r, c);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (ROWS * 64) + off) = pl[p];
    }
}

// ---- weights pre-split (B operand of the NT kernel) --------------------------------------------------------
// One tiny kernel per call splits the [N,K] weight matrix (or its transpose, for dX) into P bf16 planes
// planes[p][row][col] in a caller workspace; the GEMM blocks then stage B as plain 16-byte copies -- no VALU work for B,
// which every one of the M/128 row blocks would otherwise repeat (half of the staging VALU of a 128x128 tile).
template <int P>
__global__ void split_w_kernel(const float* __restrict__ w, int rows_in, int cols_in, int transpose, unsigned short* __restrict__ planes) {
    const int64_t total = (int64_t)rows_in * cols_in;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT [rows_out, cols_out]; transpose: out[r][c] = w[c][r]
        const int cols_out = transpose ? rows_in : cols_in;
        const int r = (int)(i / cols_out), c = (int)(i % cols_out);
        float x = transpose ? w[(int64_t)c * cols_in + r] : w[i];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const __bf16 h = (__bf16)x;                               // RNE
            const unsigned short u = __builtin_bit_cast(unsigned short, h);
            planes[(int64_t)p * total + i] = u;
            x -= __builtin_bit_cast(float, (unsigned)u << 16);
        }
    }
}

// pre-split B: this thread's chunks as P 16-byte pieces each (branch-free, clamped like split_load)
template <int ROWS, int P>
__device__ __forceinline__ void split_load_pre(const unsigned short* __restrict__ Bp, int64_t plane_stride, int ld, int row0, int nrows,
                                               int k0, int K, u32x4 (&regs)[SplitChunks<ROWS>::value][P]) {
    const int tid = threadIdx.x;
    const int last = ((nrows - row0 < ROWS) ? (nrows - row0) : ROWS) - 1;
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        int r = f >> 2;
        r = r < last ? r : last;
        int k = k0 + (f & 3) * 8;
        k = k < K - 8 ? k : K - 8;
        const unsigned off = (unsigned)(((row0 + r) * ld + k) * 2);        // bytes; N*K*2 < 2^31 for every weight matrix here
#pragma unroll
        for (int p = 0; p < P; ++p)
            regs[i][p] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Bp + p * plane_stride) + off);
    }
}

template <int ROWS, int P>
__device__ __forceinline__ void split_store_pre(unsigned char* __restrict__ S, const u32x4 (&regs)[SplitChunks<ROWS>::value][P], int k0,
                                                int nvalid, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const int r = f >> 2, c = f & 3;
        const bool valid = r < nvalid && k0 + c * 8 < K;
        const int off = split_off(r, c);
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (ROWS * 64) + off) = valid ? regs[i][p] : z;
    }
}

template <int WM, int WN, int TM, int TN, int PRODUCTS, bool BNIN, bool BNB, bool BPRE, int AMODE = 0>
__global__ __launch_bounds__(256, (BNB || (BNIN && BPRE)) ? 2 : 3) void gemm_nt_split_kernel(const float* __restrict__ A, int64_t lda, RowScale as,
                                                                       const float* __restrict__ B, int64_t ldb,
                                                                       float* __restrict__ C, int64_t ldc,
                                                                       int64_t M, int N, int K, Epilogue ep, unsigned ntn, InBN ib, int abl,
                                                                       ConvGather cg) {
    // AMODE 1 / 2: A is the im2col view of an NHWC tensor (forward / dX of a dense convolution, cg; c % 4 == 0 / c % 8 == 0)
    constexpr bool CONV = AMODE != 0;
    static_assert(!CONV || (!BNIN && !BNB && !BPRE), "the gather form has no fused BatchNorm variants");
    // abl: ablation switches of tools/gemm_bench.py (0 in production; wave-uniform kernel argument): 1 no epilogue,
    // 2 no global loads inside the K loop, 8 no staging pass (+ its barrier), 16 no fragment reads
    constexpr int P = SplitPlanes<PRODUCTS>::value;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(PRODUCTS == 8 || PRODUCTS == 6 || PRODUCTS == 3 || PRODUCTS == 1, "8 / 6 (fp32 class), 3 or 1 partial products");
    constexpr int OP_FLOATS = P * (BM + BN) * 16;                 // operand image: P planes x rows x 64 bytes
    constexpr int EP_FLOATS = WM * 32 * (BN + 4);                 // epilogue band
    constexpr int SMEM_FLOATS = OP_FLOATS > EP_FLOATS ? OP_FLOATS : EP_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    unsigned char* As = reinterpret_cast<unsigned char*>(smem);
    unsigned char* Bs = As + P * BM * 64;

    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * BM;
    const int n0 = (int)(bid % ntn) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    constexpr int NA = SplitChunks<BM>::value, NB = SplitChunks<BN>::value;
    const int mvalid = (int)((M - m0 < BM) ? (M - m0) : BM), nvalid = (N - n0 < BN) ? (N - n0) : BN;
    float4 ra[NA][2], rb[BPRE ? 1 : NB][2];
    u32x4 rbp[BPRE ? NB : 1][P];                      // BPRE: B arrives as bf16 planes (split_w_kernel), ldb = K
    const unsigned short* Bpl = reinterpret_cast<const unsigned short*>(B);
    const int64_t bstride = (int64_t)N * K;
    float sa0[NA], sa1[NA], sb0[NB], sb1[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) { sb0[i] = 1.f; sb1[i] = 1.f; }
    int rn[NA], ry[NA], rx[NA], cci[2] = {0, 0}; // CONV: this thread's rows on the row grid; channels of its two half-chunks in flight
    float ca0[CONV ? NA : 1][2], ca1[CONV ? NA : 1][2];   // CONV: plane factors per row and half-chunk
    unsigned okm = 0u;
    if constexpr (CONV) {
        split_conv_rows<BM>(cg, m0, M, rn, ry, rx);
        split_conv_load<BM, AMODE>(A, cg, rn, ry, rx, 0, K, ra, ca0, ca1, okm, cci);
    } else {
        split_row_scales<BM>(as, m0, M, sa0, sa1);
        split_load<BM>(A, lda, m0, M, 0, K, ra);
    }
    if constexpr (BPRE) split_load_pre<BN, P>(Bpl, bstride, K, n0, N, 0, K, rbp);
    else split_load<BN>(B, ldb, n0, N, 0, K, rb);
    float4 psc[2], psh[2];                       // BNIN: (scale, shift) of this thread's 8 channels of the tile in flight
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    psc[0] = psc[1] = psh[0] = psh[1] = z4;
    const int pk = (tid & 3) * 8;
    if constexpr (BNIN) {
        if (pk < K) {
            psc[0] = *reinterpret_cast<const float4*>(ib.sc + pk); psc[1] = *reinterpret_cast<const float4*>(ib.sc + pk + 4);
            psh[0] = *reinterpret_cast<const float4*>(ib.sh + pk); psh[1] = *reinterpret_cast<const float4*>(ib.sh + pk + 4);
        }
    }
    if constexpr (CONV) split_conv_store<BM, P>(As, ra, cci, cg.split, ca0, ca1, okm);
    else split_store<BM, P, true, BNIN>(As, ra, 0, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi, mvalid, K);
    if constexpr (BPRE) split_store_pre<BN, P>(Bs, rbp, 0, nvalid, K);
    else split_store<BN, P, false, false>(Bs, rb, 0, 0, sb0, sb1, psc, psh, 1.f, 0.f, nvalid, K);
    __syncthreads();

    // fragment addresses: row li of tile t, k-chunk 2s + hi; the swizzle term depends on li only
    const int swz = (li >> 2) & 3;
    const int fo0 = li * 64 + (((0 + hi) ^ swz) << 4), fo1 = li * 64 + (((2 + hi) ^ swz) << 4);
    const unsigned char* Aw = As + (wm * TM) * 32 * 64;
    const unsigned char* Bw = Bs + (wn * TN) * 32 * 64;

    const int nk = (K + SPLIT_BK - 1) / SPLIT_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more && !(abl & 2)) {  // next tile's global loads fly during the MFMA phase
            if constexpr (CONV) split_conv_load<BM, AMODE>(A, cg, rn, ry, rx, (kt + 1) * SPLIT_BK, K, ra, ca0, ca1, okm, cci);
            else split_load<BM>(A, lda, m0, M, (kt + 1) * SPLIT_BK, K, ra);
            if constexpr (BPRE) split_load_pre<BN, P>(Bpl, bstride, K, n0, N, (kt + 1) * SPLIT_BK, K, rbp);
            else split_load<BN>(B, ldb, n0, N, (kt + 1) * SPLIT_BK, K, rb);
        }
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fo = (abl & 16) ? fo0 : (s == 0 ? fo0 : fo1);
            bf16x8 b[TN][P];
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    b[u][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bw + p * (BN * 64) + u * (32 * 64) + fo));
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                bf16x8 a[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Aw + p * (BM * 64) + t * (32 * 64) + fo));
                // partial products, smallest first (accumulate chains forward the accumulator: no stall between them)
#pragma unroll
                for (int q = 0; q < PRODUCTS; ++q) {
                    const int pa = SplitTerm<PRODUCTS>::pa(q);
                    const int pb = SplitTerm<PRODUCTS>::pb(q);
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[u][pb], acc[t][u], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (BNIN) {    // (scale, shift) of the next tile's channels: L1/L2 hits, fetched behind the barrier wait
            if (more) {          // (held across the MFMA phase they cost 16 VGPRs there and made the kernel spill)
                const int k = (kt + 1) * SPLIT_BK + pk;
                psc[0] = psc[1] = psh[0] = psh[1] = z4;
                if (k < K) {
                    psc[0] = *reinterpret_cast<const float4*>(ib.sc + k); psc[1] = *reinterpret_cast<const float4*>(ib.sc + k + 4);
                    psh[0] = *reinterpret_cast<const float4*>(ib.sh + k); psh[1] = *reinterpret_cast<const float4*>(ib.sh + k + 4);
                }
            }
        }
        __syncthreads();
        if (more && !(abl & 8)) {
            if constexpr (CONV) split_conv_store<BM, P>(As, ra, cci, cg.split, ca0, ca1, okm);
            else split_store<BM, P, true, BNIN>(As, ra, (kt + 1) * SPLIT_BK, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi, mvalid, K);
            if constexpr (BPRE) split_store_pre<BN, P>(Bs, rbp, (kt + 1) * SPLIT_BK, nvalid, K);
            else split_store<BN, P, false, false>(Bs, rb, (kt + 1) * SPLIT_BK, 0, sb0, sb1, psc, psh, 1.f, 0.f, nvalid, K);
            __syncthreads();
        }
    }

    if (abl & 1) {          // ablation: keep the accumulators alive with one store per thread
        float sacc = 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[t][u][r];
        if (m0 + tid < M) C[(m0 + tid) * ldc + n0] = sacc;
        return;
    }
    nt_epilogue<WM, WN, TM, TN, AMODE, BNB, SMEM_FLOATS>(smem, acc, C, ldc, M, N, ep, cg, m0, n0, bid, ntn);
}

// ---- TN (dW): C[P,Q] = sum_m A[m,p]*sa[m] * B[m,q]*rs(m,q), both operands [m][channel] in memory ----------------
// The bf16 MFMA wants 8 consecutive k (= m) per lane, i.e. the operands TRANSPOSED: a thread loads a 4(m) x 4(channel)
// micro-tile (4 float4, channel-contiguous), splits each channel's 4 m-values and writes 8 bytes per (plane, channel) --
// half of a 16-byte [channel][8 m] atom.  LDS image per plane: atom(c, ch) = c*CH + (ch&3)*(CH/4) + ((ch>>2) ^ (4*(ch&3)))
// (c = 8-m chunk 0..3 of the 32-m stage): a 16-lane ds_write_b64 group = 8 consecutive channel quads x 2 halves covers
// all 32 banks, and the fragment reads (lane l: channel l & 31 of the wave tile, chunk 2s + (l >> 5)) hit 16 distinct
// 16-byte slots per 16-lane group -- both conflict free, no padding (48 KB per block, 3 blocks per CU).
// Thread -> micro-tile: cq = (lane & 7) | ((lane >> 4) << 3), mq = ((lane >> 3) & 1) | (wave << 1): one global load
// instruction reads 2 rows x 4 full 128-byte lines per wave.
template <int CH>
__device__ __forceinline__ int tn_atom(int c, int ch) {
    // (32 channels: 8 atoms per residue class, the swizzle stays inside them; stores and fragment reads conflict-free like the wider forms)
    if (CH == 32) return c * CH + (ch & 3) * 8 + ((ch >> 2) ^ (2 * (ch & 3)));
    return c * CH + (ch & 3) * (CH / 4) + ((ch >> 2) ^ (4 * (ch & 3)));
}

template <int CH>
__device__ __forceinline__ void tn_split_load(const float* __restrict__ Pm, int64_t ld, int64_t m0, int64_t mend, int c0, int ncols,
                                              int cq, int mq, float4 (&regs)[4]) {
    // branch-free like split_load: rows past the chunk end / columns past the matrix are clamped, zeroed by the store pass
    const char* __restrict__ base = reinterpret_cast<const char*>(Pm + m0 * ld);
    const int last = (int)((mend - m0 < 32) ? (mend - m0) : 32) - 1;
    int c = c0 + cq * 4;
    c = c < ncols - 4 ? c : ncols - 4;                                  // ncols % 4 == 0, ncols >= 4
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        int r = mq * 4 + jj;
        r = r < last ? r : last;
        regs[jj] = *reinterpret_cast<const float4*>(base + (unsigned)((r * (int)ld + c) * 4));
    }
}

// per-row factors of this thread's 4 rows (small planes, L2 hits): fetched AFTER the MFMA phase, behind the barrier wait
// (held across the MFMA phase their 16 VGPRs made the kernel spill)
__device__ __forceinline__ void tn_split_factors(int64_t m0, int64_t mend, const float* __restrict__ rowmul, const RowScale& rs, int mq,
                                                 float (&f0)[4], float (&f1)[4]) {
    const int last = (int)((mend - m0 < 32) ? (mend - m0) : 32) - 1;
    int r[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { r[jj] = mq * 4 + jj; r[jj] = r[jj] < last ? r[jj] : last; }    // clamped: rows past the end are zeroed by the store
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { f0[jj] = 1.f; f1[jj] = 1.f; }
    if (rowmul != nullptr) {            // wave-uniform branches only; the loads inside issue back to back
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { f0[jj] = rowmul[m0 + r[jj]]; f1[jj] = f0[jj]; }
    }
    if (rs.r0 != nullptr) {
        float t0[4], t1[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) t0[jj] = rs.r0[m0 + r[jj]];
        if (rs.r1 != nullptr) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) t1[jj] = rs.r1[m0 + r[jj]];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { f0[jj] *= t0[jj]; f1[jj] *= t1[jj]; }
    }
}

// The staging of one channel e of a thread's 4(m) x 4(channel) micro-tile in three pieces (the pipelined kernel places them
// between its MFMAs one at a time; tn_split_store below runs them back to back):
//   tn_quad_prepare: the 4 m-values -> BatchNorm + activation (BNIN), row factor, zero outside the matrix
//   tn_quad_split:   -> P planes, 2 dwords each (m 0,1 | m 2,3)
//   tn_quad_write:   8 bytes per plane into the [channel][8 m] atoms
template <bool BNIN>
__device__ __forceinline__ void tn_quad_prepare(float (&x)[4], const float4 (&regs)[4], int e, int cbase, int split, bool split_active,
                                                const float (&f0)[4], const float (&f1)[4], int mq, float sce, float she, float neg, float hi,
                                                int left, bool col_ok, int conv_ci, unsigned conv_ok) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        x[jj] = e == 0 ? regs[jj].x : e == 1 ? regs[jj].y : e == 2 ? regs[jj].z : regs[jj].w;
        if constexpr (BNIN) x[jj] = bn_act_load(x[jj], sce, she, neg, hi);
        x[jj] *= (!split_active || cbase + e < split) ? f0[jj] : f1[jj];
        const bool row_ok = conv_ci >= 0 ? ((conv_ok >> jj) & 1u) != 0u : mq * 4 + jj < left;
        x[jj] = (col_ok && row_ok) ? x[jj] : 0.f;                       // clamped loads: outside the matrix = 0 (select: NaN safe)
    }
}
template <int P>
__device__ __forceinline__ void tn_quad_split(const float (&x)[4], unsigned (&w0)[P], unsigned (&w1)[P]) {
    f32x2 lo = {x[0], x[1]}, hi2 = {x[2], x[3]};
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bf16x2 h0 = __builtin_convertvector(lo, bf16x2), h1 = __builtin_convertvector(hi2, bf16x2);
        const unsigned u0 = __builtin_bit_cast(unsigned, h0), u1 = __builtin_bit_cast(unsigned, h1);
        w0[p] = u0; w1[p] = u1;
        if (p + 1 < P) {
            lo = lo - f32x2{__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xffff0000u)};
            hi2 = hi2 - f32x2{__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xffff0000u)};
        }
    }
}
template <int CH, int P>
__device__ __forceinline__ void tn_quad_write(unsigned char* __restrict__ S, int e, int cq, int mq, const unsigned (&w0)[P], const unsigned (&w1)[P]) {
    const int off = tn_atom<CH>(mq >> 1, cq * 4 + e) * 16 + (mq & 1) * 8;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(S + p * (CH * 64) + off) = u32x2{w0[p], w1[p]};
    }
}

template <int CH, int P, bool BNIN>
__device__ __forceinline__ void tn_split_store(unsigned char* __restrict__ S, const float4 (&regs)[4], int c0, int split, bool split_active,
                                               const float (&f0)[4], const float (&f1)[4], int cq, int mq,
                                               const float4 sc, const float4 sh, float neg, float hi, int left, int ncols,
                                               int conv_ci = -1, unsigned conv_ok = 0u) {
    // conv_ci >= 0 (gathered B of a dense conv's dW): the split index is the source channel, row validity comes as a bit mask
    if (CH != 128 && cq >= CH / 4) return;
    const bool col_ok = c0 + cq * 4 < ncols;
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const int cbase = conv_ci >= 0 ? conv_ci : c0 + cq * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x[4];
        unsigned w0[P], w1[P];
        tn_quad_prepare<BNIN>(x, regs, e, cbase, split, split_active, f0, f1, mq, scv[e], shv[e], neg, hi, left, col_ok, conv_ci, conv_ok);
        tn_quad_split<P>(x, w0, w1);
        tn_quad_write<CH, P>(S, e, cq, mq, w0, w1);
    }
}

// gathered B operand (dW of a dense conv): rows m = output pixels, this thread's column quad = one (tap, 4 channels) for the
// whole tile.  Branch-free: rows past the chunk / taps outside the image read pixel 0 and are zeroed by the store (okm).
// spx: the 4 source pixels (32-bit), kept for the plane factors fetched after the MFMA phase.
__device__ __forceinline__ void tn_conv_split_load(const float* __restrict__ src, const ConvGather& cg, int64_t m0, int64_t mend, int ky, int kx,
                                                   int ci, bool kok, int mq, float4 (&regs)[4], int (&spx)[4], unsigned& okm) {
    okm = 0u;
    const int last = (int)(mend - 1);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        int row = (int)m0 + mq * 4 + jj;
        const bool in = row <= last;
        row = in ? row : last;
        const int q = row / cg.rw, rx = row - q * cg.rw;
        const int n = q / cg.rh, ry = q - n * cg.rh;
        int sy = 0, sx = 0;
        const bool ok = in && kok && conv_src<1>(cg, ry, rx, ky, kx, sy, sx);
        const int spix = ok ? (n * cg.h + sy) * cg.w + sx : 0;
        regs[jj] = *reinterpret_cast<const float4*>(src + (int64_t)spix * cg.c + ci);
        spx[jj] = spix;
        okm |= ok ? (1u << jj) : 0u;
    }
}
__device__ __forceinline__ void tn_conv_factors(const ConvGather& cg, const int (&spx)[4], float (&f0)[4], float (&f1)[4]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { f0[jj] = 1.f; f1[jj] = 1.f; }
    if (cg.p0 != nullptr) {                       // wave-uniform branches; the loads inside issue back to back
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) f0[jj] = cg.p0[spx[jj]];
        if (cg.p1 != nullptr) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) f1[jj] = cg.p1[spx[jj]];
        }
    }
}

// (Round 3 measured a software-pipelined form of this kernel -- one block per CU, two LDS buffers, the staging of stage s+1 in
// slices between the MFMA pairs of stage s, one LDS-only barrier per stage; commit e9fb585, gemm_tn_pipe_kernel: exact, and
// 135 vs 175 TF/s on the 65536 x 1024 x 1024 dW, slower on every shape of tools/gemm_bench.py.  With one wave per SIMD every
// fragment-read or wait stall idles the matrix pipe; three co-resident blocks in alternating phases hide more.)
template <int WM, int WN, int TM, int TN, int PRODUCTS, bool BNIN, bool BCONV = false>
__global__ __launch_bounds__(256, 3) void gemm_tn_split_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ sa,
                                                            const float* __restrict__ B, int64_t ldb, RowScale sb,
                                                            float* __restrict__ Cws, int64_t M, int Pn, int Q, int64_t chunk, InBN ib,
                                                            unsigned qtiles, unsigned ptiles, ConvGather cg) {
    constexpr int P = SplitPlanes<PRODUCTS>::value;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(!BCONV || !BNIN, "the gathered B operand has no BatchNorm-on-load form");
    static_assert((BM == 128 || BM == 64 || BM == 32) && (BN == 128 || BN == 64), "micro-tile mapping: 32 (16, 8) channel quads x 8 m quads");
    __shared__ __attribute__((aligned(16))) unsigned char smem[P * (BM + BN) * 64];
    unsigned char* As = smem;
    unsigned char* Bs = smem + P * BM * 64;

    // 1-D grid, XCD-aware: the (p, q) tiles of one m-chunk get consecutive logical ids = one XCD, so the tiles that re-read
    // the same A / B panels hit that XCD's L2 (the 3-D grid dealt them round-robin to the 8 L2s: 3x HBM re-reads measured)
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned zsplit = bid / (qtiles * ptiles), rem = bid % (qtiles * ptiles);
    const int q0 = (int)(rem % qtiles) * BN, p0 = (int)(rem / qtiles) * BM;
    const int64_t mbeg = (int64_t)zsplit * chunk;
    const int64_t mend = (mbeg + chunk < M) ? mbeg + chunk : M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;
    const int cq = (lane & 7) | ((lane >> 4) << 3), mq = ((lane >> 3) & 1) | (wave << 1);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    float4 ra[4], rb[4];
    float fa0[4], fa1[4], fb0[4], fb1[4];
    const RowScale none = {nullptr, nullptr, 0};
    const bool sb_active = sb.r0 != nullptr;
    float4 bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsh = bsc;     // BNIN: (scale, shift) of this thread's 4 B columns, fixed for the block
    if constexpr (BNIN) {
        const int q = q0 + cq * 4;
        if ((BN == 128 || cq < BN / 4) && q < Q) { bsc = *reinterpret_cast<const float4*>(ib.sc + q); bsh = *reinterpret_cast<const float4*>(ib.sh + q); }
    }
    // BCONV: this thread's B column quad = (tap ky, kx; channels cci .. cci + 3) of the source tensor, fixed for the block
    int cky = 0, ckx = 0, cci = -1, spx[4] = {0, 0, 0, 0};
    bool ckok = false;
    unsigned okm = 0u;
    if constexpr (BCONV) {
        int k = q0 + cq * 4;
        ckok = k < Q;
        k = ckok ? k : Q - 4;
        const int t = k / cg.c;
        cci = k - t * cg.c;
        cky = t / cg.kw; ckx = t - cky * cg.kw;
    }
    tn_split_load<BM>(A, lda, mbeg, mend, p0, Pn, cq, mq, ra);
    if constexpr (BCONV) tn_conv_split_load(B, cg, mbeg, mend, cky, ckx, cci, ckok, mq, rb, spx, okm);
    else tn_split_load<BN>(B, ldb, mbeg, mend, q0, Q, cq, mq, rb);
    tn_split_factors(mbeg, mend, sa, none, mq, fa0, fa1);
    if constexpr (BCONV) tn_conv_factors(cg, spx, fb0, fb1);
    else tn_split_factors(mbeg, mend, nullptr, sb, mq, fb0, fb1);
    int left = (int)((mend - mbeg < 32) ? (mend - mbeg) : 32);
    tn_split_store<BM, P, false>(As, ra, p0, 0, false, fa0, fa1, cq, mq, bsc, bsh, 1.f, 0.f, left, Pn);
    if constexpr (BCONV) tn_split_store<BN, P, false>(Bs, rb, q0, cg.split, true, fb0, fb1, cq, mq, bsc, bsh, 1.f, 0.f, left, Q, cci, okm);
    else tn_split_store<BN, P, BNIN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1, cq, mq, bsc, bsh, ib.neg, ib.hi, left, Q);
    __syncthreads();

    // fragment byte offsets inside a plane: channel li of wave tile t (tile base multiple of 32), chunk 2s + hi
    int foA[TM][2], foB[TN][2];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) foA[t][s] = tn_atom<BM>(2 * s + hi, (wm * TM + t) * 32 + li) * 16;
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s) foB[u][s] = tn_atom<BN>(2 * s + hi, (wn * TN + u) * 32 + li) * 16;

    // one MFMA phase over the 32-m stage in LDS
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b[TN][P];
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    b[u][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + p * (BN * 64) + foB[u][s]));
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                bf16x8 a[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + p * (BM * 64) + foA[t][s]));
#pragma unroll
                for (int q = 0; q < PRODUCTS; ++q) {
                    const int pa = SplitTerm<PRODUCTS>::pa(q);
                    const int pb = SplitTerm<PRODUCTS>::pb(q);
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[u][pb], acc[t][u], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // every iteration loads the NEXT stage unconditionally (the last stage is peeled): a conditional load made hipcc
    // merge register copies behind the loads and wait for them before the MFMA phase
    for (int64_t mt = mbeg; mt + SPLIT_BK < mend; mt += SPLIT_BK) {
        tn_split_load<BM>(A, lda, mt + SPLIT_BK, mend, p0, Pn, cq, mq, ra);
        if constexpr (BCONV) tn_conv_split_load(B, cg, mt + SPLIT_BK, mend, cky, ckx, cci, ckok, mq, rb, spx, okm);
        else tn_split_load<BN>(B, ldb, mt + SPLIT_BK, mend, q0, Q, cq, mq, rb);
        mfma_phase();
        tn_split_factors(mt + SPLIT_BK, mend, sa, none, mq, fa0, fa1);
        if constexpr (BCONV) tn_conv_factors(cg, spx, fb0, fb1);
        else tn_split_factors(mt + SPLIT_BK, mend, nullptr, sb, mq, fb0, fb1);
        __syncthreads();
        left = (int)((mend - mt - SPLIT_BK < 32) ? (mend - mt - SPLIT_BK) : 32);
        tn_split_store<BM, P, false>(As, ra, p0, 0, false, fa0, fa1, cq, mq, bsc, bsh, 1.f, 0.f, left, Pn);
        if constexpr (BCONV) tn_split_store<BN, P, false>(Bs, rb, q0, cg.split, true, fb0, fb1, cq, mq, bsc, bsh, 1.f, 0.f, left, Q, cci, okm);
        else tn_split_store<BN, P, BNIN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1, cq, mq, bsc, bsh, ib.neg, ib.hi, left, Q);
        __syncthreads();
    }
    mfma_phase();

    float* Cz = Cws + (int64_t)zsplit * Pn * Q;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + (wm * TM + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (p >= Pn) continue;
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int q = q0 + (wn * TN + u) * 32 + li;
                if (q < Q) Cz[(int64_t)p * Q + q] = acc[t][u][r];
            }
        }
}

// ---- mode switch --------------------------------------------------------------------------------------------
// 0: f32-input MFMA (gemm.hip) | 6 (default): split-bf16, fp32 class | 3: split-bf16, 2 planes (opt-in)
static int env_products() {
    const char* e = getenv("TSII_GEMM_PRODUCTS");
    if (e == nullptr) return 6;
    const int v = atoi(e);
    return (v == 0 || v == 1 || v == 3 || v == 6 || v == 8) ? v : 6;
}
static thread_local int g_products = env_products();      // per calling thread (tsii_set_gemm_products)
// (TSII_GEMM_PRODUCTS is the documented process default of the arithmetic switch: include/tsii_hip.h.  The two knobs below exist in
// A/B builds only -- -DTSII_GEMM_PC_ABLATIONS, tools/variants/build_variant.py; the stock library reads nothing else.)
#ifdef TSII_GEMM_PC_ABLATIONS
static int g_abl = getenv("TSII_GEMM_ABL") ? atoi(getenv("TSII_GEMM_ABL")) : 0;     // tools/gemm_bench.py ablations
static const int g_force_tile = getenv("TSII_GEMM_TILE") ? atoi(getenv("TSII_GEMM_TILE")) : 0;   // 1 = 128x64 tiles everywhere
#else
static constexpr int g_abl = 0, g_force_tile = 0;
#endif

int gemm_products() { return g_products; }

static const ConvGather kNoGather = {0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0, 0, 0};

template <int WM, int WN, int TM, int TN, int PRODUCTS>
static int launch_nt_split_cfg(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t M, int N, int K, Epilogue ep, InBN ib, bool bpre, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "gemm_nt_split: grid too large");
    if (ep.bn_y != nullptr) {
        TSII_REQUIRE(ib.sc == nullptr && N % 4 == 0 && ldc == N && aligned16(ep.bn_y) && ep.vec_store,
                     "gemm_nt_split: the BatchNorm-backward epilogue needs N %% 4 == 0 and 16-byte aligned operands");
        if (bpre) hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, true, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
        else hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, true, false>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
    } else if (ib.sc != nullptr) {
        TSII_REQUIRE(aligned16(ib.sc) && aligned16(ib.sh), "gemm_nt_split: input BatchNorm needs 16-byte aligned scale / shift");
        if (bpre) hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, true, false, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
        else hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, true, false, false>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
    } else {
        if (bpre) hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, false, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
        else hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, false, false>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib, g_abl, kNoGather);
    }
    return check_launch("gemm_nt_split");
}

bool nt_split_ok(const float* A, int64_t lda, const float* B, int64_t ldb, int K) {
    return g_products != 0 && K % 8 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B);
}

size_t nt_split_ws_bytes(int n, int k) { return (size_t)3 * n * k * sizeof(unsigned short) + 16; }   // 3 planes (fewer in modes 1 / 3)

// B = fp32 [N,K] (b_transposed: fp32 [K,N], i.e. the weight as stored, for dX).  With a workspace of nt_split_ws_bytes(N, K)
// the weights are split into bf16 planes once (split_w_kernel) and every block stages them as plain copies; without one
// (wsplit == nullptr) the blocks split B while staging (needs B as [N,K]).
int launch_nt_split(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, bool b_transposed, float* C, int64_t ldc,
                    int64_t M, int N, int K, Epilogue ep, InBN ib, void* wsplit, hipStream_t stream) {
    ep.vec_store = (ldc % 4 == 0) && aligned16(C);
    bool bpre = false;
    if (wsplit != nullptr && (int64_t)N * K * 2 < (1ll << 31)) {
        unsigned short* planes = reinterpret_cast<unsigned short*>((reinterpret_cast<uintptr_t>(wsplit) + 15) & ~(uintptr_t)15);
        const int rows_in = b_transposed ? K : N, cols_in = b_transposed ? N : K;
        const unsigned g = stream_grid((int64_t)N * K, 256);
        if (g_products == 1) hipLaunchKernelGGL(split_w_kernel<1>, dim3(g), dim3(256), 0, stream, B, rows_in, cols_in, b_transposed ? 1 : 0, planes);
        else if (g_products == 3) hipLaunchKernelGGL(split_w_kernel<2>, dim3(g), dim3(256), 0, stream, B, rows_in, cols_in, b_transposed ? 1 : 0, planes);
        else hipLaunchKernelGGL(split_w_kernel<3>, dim3(g), dim3(256), 0, stream, B, rows_in, cols_in, b_transposed ? 1 : 0, planes);
        int rc = check_launch("split_w");
        if (rc) return rc;
        B = reinterpret_cast<const float*>(planes);
        ldb = K;
        bpre = true;
    } else {
        TSII_REQUIRE(!b_transposed, "gemm_nt_split: a transposed B needs the split workspace");
    }
#define TSII_NT_SPLIT(WM, WN, TM, TN) \
    (g_products == 1 ? launch_nt_split_cfg<WM, WN, TM, TN, 1>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, bpre, stream) \
     : g_products == 8 ? launch_nt_split_cfg<WM, WN, TM, TN, 8>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, bpre, stream) \
     : g_products == 3 ? launch_nt_split_cfg<WM, WN, TM, TN, 3>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, bpre, stream) \
                       : launch_nt_split_cfg<WM, WN, TM, TN, 6>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, bpre, stream))
    if (g_force_tile == 1 && N > 32) return TSII_NT_SPLIT(2, 2, 2, 1);
    if (N % 128 == 0 || N > 192) return TSII_NT_SPLIT(2, 2, 2, 2);
    if (N > 32) return TSII_NT_SPLIT(2, 2, 2, 1);
    return TSII_NT_SPLIT(4, 1, 1, 1);
#undef TSII_NT_SPLIT
}

// ---- dense convolutions: A gathered (AMODE 1 forward, 2 dX), B = the fp32 [N,K] weight layout, split while staged ----
bool nt_split_conv_ok(int amode, const float* A, const float* B, int64_t ldb, int K, const ConvGather& cg) {
    return g_products != 0 && cg.c % (amode == 2 ? 8 : 4) == 0 && K % 8 == 0 && ldb % 4 == 0 && cg.pfull == nullptr && aligned16(A) && aligned16(B);
}

template <int WM, int WN, int TM, int TN, int AMODE>
static int launch_nt_split_conv_cfg(const float* A, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                                    Epilogue ep, const ConvGather& cg, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "conv gemm (split): grid too large");
    const RowScale none = {nullptr, nullptr, 0};
    const InBN nobn = {nullptr, nullptr, 1.f, 0.f};
#define TSII_NT_CONV(PR) hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PR, false, false, false, AMODE>), dim3((unsigned)nblocks), dim3(256), 0, stream, \
                                            A, (int64_t)0, none, B, ldb, C, ldc, M, N, K, ep, ntn, nobn, 0, cg)
    if (g_products == 1) TSII_NT_CONV(1);
    else if (g_products == 3) TSII_NT_CONV(3);
    else TSII_NT_CONV(6);                        // 8 products: not instantiated for the gather form, the 6-product form is fp32 class
#undef TSII_NT_CONV
    return check_launch("conv_gemm_nt_split");
}

int launch_nt_split_conv(int amode, const float* A, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                         Epilogue ep, const ConvGather& cg, hipStream_t stream) {
    ep.vec_store = (ldc % 4 == 0) && aligned16(C);
#define TSII_NT_CONV_T(WM, WN, TM, TN) \
    (amode == 1 ? launch_nt_split_conv_cfg<WM, WN, TM, TN, 1>(A, B, ldb, C, ldc, M, N, K, ep, cg, stream) \
                : launch_nt_split_conv_cfg<WM, WN, TM, TN, 2>(A, B, ldb, C, ldc, M, N, K, ep, cg, stream))
    if (N % 128 == 0 || N > 192) return TSII_NT_CONV_T(2, 2, 2, 2);
    if (N > 32) return TSII_NT_CONV_T(2, 2, 2, 1);
    return TSII_NT_CONV_T(4, 1, 1, 1);
#undef TSII_NT_CONV_T
}

bool tn_split_ok(const float* A, int64_t lda, const float* B, int64_t ldb, int Pn, int Q) {
    return g_products != 0 && Pn % 4 == 0 && Q % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B);
}

// tile = 0: 128x128, 1: 128x64 (narrow), 2: 64x64 (P or Q below 128), 3: 32x128 (P <= 32, one wave row); one partial [P,Q] slab per m-chunk in Cws
int launch_tn_split(const float* A, int64_t lda, const float* sa, const float* B, int64_t ldb, RowScale sb, float* Cws,
                    int64_t M, int Pn, int Q, int64_t chunk, int splits, int tile, InBN ib, hipStream_t stream) {
    const int bm = tile == 3 ? 32 : (tile == 2 ? 64 : 128), bn = (tile == 0 || tile == 3) ? 128 : 64;
    const unsigned qt = (unsigned)cdiv(Q, bn), pt = (unsigned)cdiv(Pn, bm);
    const int64_t nblocks = (int64_t)qt * pt * splits;
    TSII_REQUIRE(nblocks < (1ll << 31), "gemm_tn_split: grid too large");
    const dim3 grid((unsigned)nblocks);
    if (ib.sc != nullptr) TSII_REQUIRE(aligned16(ib.sc) && aligned16(ib.sh), "gemm_tn_split: input BatchNorm needs 16-byte aligned scale / shift");
#define TSII_TN_SPLIT(TMV, TNV, PR, BNV) hipLaunchKernelGGL((gemm_tn_split_kernel<2, 2, TMV, TNV, PR, BNV>), grid, dim3(256), 0, stream, A, lda, sa, B, ldb, sb, Cws, M, Pn, Q, chunk, ib, qt, pt, kNoGather)
#define TSII_TN_SPLIT_THIN(PR, BNV) hipLaunchKernelGGL((gemm_tn_split_kernel<1, 4, 1, 1, PR, BNV>), grid, dim3(256), 0, stream, A, lda, sa, B, ldb, sb, Cws, M, Pn, Q, chunk, ib, qt, pt, kNoGather)
#define TSII_TN_SPLIT_T(PR, BNV) do { if (tile == 0) TSII_TN_SPLIT(2, 2, PR, BNV); else if (tile == 1) TSII_TN_SPLIT(2, 1, PR, BNV); else if (tile == 3) TSII_TN_SPLIT_THIN(PR, BNV); else TSII_TN_SPLIT(1, 1, PR, BNV); } while (0)
    if (ib.sc != nullptr) { if (g_products == 1) TSII_TN_SPLIT_T(1, true); else if (g_products == 8) TSII_TN_SPLIT_T(8, true); else if (g_products == 3) TSII_TN_SPLIT_T(3, true); else TSII_TN_SPLIT_T(6, true); }
    else { if (g_products == 1) TSII_TN_SPLIT_T(1, false); else if (g_products == 8) TSII_TN_SPLIT_T(8, false); else if (g_products == 3) TSII_TN_SPLIT_T(3, false); else TSII_TN_SPLIT_T(6, false); }
#undef TSII_TN_SPLIT_T
#undef TSII_TN_SPLIT
    return check_launch("gemm_tn_split");
}

// dW of a dense convolution: A = dy [M, Pn = cout] (x sa = 1/count), B = im2col view of x by cg (Q = taps * cin, cin % 4 == 0)
bool tn_split_conv_ok(const float* A, int64_t lda, const float* B, int64_t M, int Pn, int Q, const ConvGather& cg) {
    return g_products != 0 && Pn % 4 == 0 && Q % 4 == 0 && cg.c % 4 == 0 && lda % 4 == 0 && cg.pfull == nullptr && aligned16(A) && aligned16(B) &&
           M < (1ll << 31) && (M / ((int64_t)cg.rh * cg.rw) + 1) * cg.h * cg.w < (1ll << 31);      // 32-bit row / source-pixel indices
}

int launch_tn_split_conv(const float* A, int64_t lda, const float* sa, const float* B, const ConvGather& cg, float* Cws, int64_t M, int Pn, int Q,
                         int64_t chunk, int splits, bool big, hipStream_t stream) {
    const int bm = big ? 128 : 64, bn = bm;
    const unsigned qt = (unsigned)cdiv(Q, bn), pt = (unsigned)cdiv(Pn, bm);
    const int64_t nblocks = (int64_t)qt * pt * splits;
    TSII_REQUIRE(nblocks < (1ll << 31), "conv gemm_tn_split: grid too large");
    const dim3 grid((unsigned)nblocks);
    const RowScale none = {nullptr, nullptr, 0};
    const InBN nobn = {nullptr, nullptr, 1.f, 0.f};
#define TSII_TN_CONV(TMV, PR) hipLaunchKernelGGL((gemm_tn_split_kernel<2, 2, TMV, TMV, PR, false, true>), grid, dim3(256), 0, stream, A, lda, sa, B, (int64_t)0, none, \
                                                 Cws, M, Pn, Q, chunk, nobn, qt, pt, cg)
    TSII_REQUIRE(big || g_products == 1, "conv gemm_tn_split: 64x64 tiles exist in the plain-bf16 mode only (the caller keeps the f32 kernel)");
    if (g_products == 1) { if (big) TSII_TN_CONV(2, 1); else TSII_TN_CONV(1, 1); }
    else if (g_products == 3) TSII_TN_CONV(2, 3);
    else TSII_TN_CONV(2, 6);
#undef TSII_TN_CONV
    return check_launch("conv_gemm_tn_split");
}

}  // namespace tsii

extern "C" int tsii_set_gemm_products(int products) {
    TSII_REQUIRE(products == -1 || products == 0 || products == 1 || products == 3 || products == 6 || products == 8,
                 "set_gemm_products: 0 (f32 MFMA), 1 (bf16 operands), 3, 6 or 8 (split bf16), -1 (back to the process default)");
    tsii::g_products = products < 0 ? tsii::env_products() : products;
    return 0;
}

extern "C" int tsii_get_gemm_products(void) { return tsii::g_products; }
