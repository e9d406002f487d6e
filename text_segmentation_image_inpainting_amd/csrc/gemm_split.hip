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

namespace tsii {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int SPLIT_BK = 32;          // K elements per tile; one LDS row = 32 bf16 = 64 bytes = 4 chunks of 8

// 8 consecutive fp32 values -> P planes of 8 bf16 (element i of a plane in bits [16*(i&1), +16) of dword i >> 1)
template <int P>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&pl)[P]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 x = {v[2 * j], v[2 * j + 1]};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bf16x2 h = __builtin_convertvector(x, bf16x2);      // v_cvt_pk_bf16_f32 (RNE)
            const unsigned u = __builtin_bit_cast(unsigned, h);
            pl[p][j] = u;
            if (p + 1 < P) {
                const f32x2 back = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
                x = x - back;                                          // exact: the low significand bits
            }
        }
    }
}

// byte offset of (row, chunk) inside one plane of an operand tile
__device__ __forceinline__ int split_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// chunks of a ROWS x 32 tile handled per thread: f = tid + 256*i -> row f >> 2, chunk f & 3 (= tid & 3 for every i)
template <int ROWS>
struct SplitChunks { static constexpr int value = (ROWS * 4 + 255) / 256; };

// raw loads of this thread's chunks: 8 consecutive k of a row-major [rows, K] fp32 matrix as 2 float4
template <int ROWS>
__device__ __forceinline__ void split_load(const float* __restrict__ Pm, int64_t ld, int64_t row0, int64_t nrows, int k0, int K,
                                           float4 (&regs)[SplitChunks<ROWS>::value][2]) {
    // wave-uniform 64-bit base (SGPRs) + 32-bit per-thread offsets: per-chunk 64-bit address VGPRs made the kernel spill
    const int tid = threadIdx.x;
    const float* __restrict__ base = Pm + row0 * ld;
    const int left = (int)((nrows - row0 < ROWS) ? (nrows - row0) : ROWS);
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        const int r = f >> 2, c = f & 3;
        const int k = k0 + c * 8;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (r < left && k < K) {     // K % 8 == 0: a chunk is all in or all out (r < ROWS whenever r < left)
            const float* p = base + (unsigned)(r * (int)ld + k);
            v0 = *reinterpret_cast<const float4*>(p);
            v1 = *reinterpret_cast<const float4*>(p + 4);
        }
        regs[i][0] = v0; regs[i][1] = v1;      // not touched until the store pass: the loads stay in flight over the MFMAs
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
                                            const float4 (&psc)[2], const float4 (&psh)[2], float neg, float hi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < SplitChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const int r = f >> 2, c = f & 3;
        float v[8] = {regs[i][0].x, regs[i][0].y, regs[i][0].z, regs[i][0].w, regs[i][1].x, regs[i][1].y, regs[i][1].z, regs[i][1].w};
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
        u32x4 pl[P];
        split8<P>(v, pl);
        const int off = split_off(r, c);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (ROWS * 64) + off) = pl[p];
    }
}

// PRODUCTS = 6 -> 3 planes, 3 -> 2 planes
template <int PRODUCTS>
struct SplitPlanes { static constexpr int value = PRODUCTS == 6 ? 3 : 2; };

template <int WM, int WN, int TM, int TN, int PRODUCTS, bool BNIN, bool BNB>
__global__ __launch_bounds__(256, BNB ? 2 : 3) void gemm_nt_split_kernel(const float* __restrict__ A, int64_t lda, RowScale as,
                                                                       const float* __restrict__ B, int64_t ldb,
                                                                       float* __restrict__ C, int64_t ldc,
                                                                       int64_t M, int N, int K, Epilogue ep, unsigned ntn, InBN ib) {
    constexpr int P = SplitPlanes<PRODUCTS>::value;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(PRODUCTS == 6 || PRODUCTS == 3, "6 (fp32 class) or 3 partial products");
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
    float4 ra[NA][2], rb[NB][2];
    float sa0[NA], sa1[NA], sb0[NB], sb1[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) { sb0[i] = 1.f; sb1[i] = 1.f; }
    split_row_scales<BM>(as, m0, M, sa0, sa1);
    split_load<BM>(A, lda, m0, M, 0, K, ra);
    split_load<BN>(B, ldb, n0, N, 0, K, rb);
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
    split_store<BM, P, true, BNIN>(As, ra, 0, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi);
    split_store<BN, P, false, false>(Bs, rb, 0, 0, sb0, sb1, psc, psh, 1.f, 0.f);
    __syncthreads();

    // fragment addresses: row li of tile t, k-chunk 2s + hi; the swizzle term depends on li only
    const int swz = (li >> 2) & 3;
    const int fo0 = li * 64 + (((0 + hi) ^ swz) << 4), fo1 = li * 64 + (((2 + hi) ^ swz) << 4);
    const unsigned char* Aw = As + (wm * TM) * 32 * 64;
    const unsigned char* Bw = Bs + (wn * TN) * 32 * 64;

    const int nk = (K + SPLIT_BK - 1) / SPLIT_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {  // next tile's global loads fly during the MFMA phase
            split_load<BM>(A, lda, m0, M, (kt + 1) * SPLIT_BK, K, ra);
            split_load<BN>(B, ldb, n0, N, (kt + 1) * SPLIT_BK, K, rb);
        }
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fo = s == 0 ? fo0 : fo1;
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
                    // PRODUCTS 6: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0);  PRODUCTS 3: (1,0) (0,1) (0,0)
                    const int pa = PRODUCTS == 6 ? (q == 0 ? 2 : (q == 1 || q == 3) ? 1 : 0) : (q == 0 ? 1 : 0);
                    const int pb = PRODUCTS == 6 ? (q == 2 ? 2 : (q == 1 || q == 4) ? 1 : 0) : (q == 1 ? 1 : 0);
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
        if (more) {
            split_store<BM, P, true, BNIN>(As, ra, (kt + 1) * SPLIT_BK, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi);
            split_store<BN, P, false, false>(Bs, rb, 0, 0, sb0, sb1, psc, psh, 1.f, 0.f);
            __syncthreads();
        }
    }

    const ConvGather nocg = {0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0, 0, 0};
    nt_epilogue<WM, WN, TM, TN, 0, BNB, SMEM_FLOATS>(smem, acc, C, ldc, M, N, ep, nocg, m0, n0, bid, ntn);
}

// ---- mode switch --------------------------------------------------------------------------------------------
// 0: f32-input MFMA (gemm.hip) | 6 (default): split-bf16, fp32 class | 3: split-bf16, 2 planes (opt-in)
static int env_products() {
    const char* e = getenv("TSII_GEMM_PRODUCTS");
    if (e == nullptr) return 6;
    const int v = atoi(e);
    return (v == 0 || v == 3 || v == 6) ? v : 6;
}
static int g_products = env_products();

int gemm_products() { return g_products; }

template <int WM, int WN, int TM, int TN, int PRODUCTS>
static int launch_nt_split_cfg(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t M, int N, int K, Epilogue ep, InBN ib, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "gemm_nt_split: grid too large");
    if (ep.bn_y != nullptr) {
        TSII_REQUIRE(ib.sc == nullptr && N % 4 == 0 && ldc == N && aligned16(ep.bn_y) && ep.vec_store,
                     "gemm_nt_split: the BatchNorm-backward epilogue needs N %% 4 == 0 and 16-byte aligned operands");
        hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib);
    } else if (ib.sc != nullptr) {
        TSII_REQUIRE(aligned16(ib.sc) && aligned16(ib.sh), "gemm_nt_split: input BatchNorm needs 16-byte aligned scale / shift");
        hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, true, false>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib);
    } else {
        hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, PRODUCTS, false, false>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, ib);
    }
    return check_launch("gemm_nt_split");
}

bool nt_split_ok(const float* A, int64_t lda, const float* B, int64_t ldb, int K) {
    return g_products != 0 && K % 8 == 0 && lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B);
}

int launch_nt_split(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, float* C, int64_t ldc,
                    int64_t M, int N, int K, Epilogue ep, InBN ib, hipStream_t stream) {
    ep.vec_store = (ldc % 4 == 0) && aligned16(C);
    const bool six = g_products != 3;
    if (N % 128 == 0 || N > 192)
        return six ? launch_nt_split_cfg<2, 2, 2, 2, 6>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream)
                   : launch_nt_split_cfg<2, 2, 2, 2, 3>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream);
    if (N > 32)
        return six ? launch_nt_split_cfg<2, 2, 2, 1, 6>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream)
                   : launch_nt_split_cfg<2, 2, 2, 1, 3>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream);
    return six ? launch_nt_split_cfg<4, 1, 1, 1, 6>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream)
               : launch_nt_split_cfg<4, 1, 1, 1, 3>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, ib, stream);
}

}  // namespace tsii

extern "C" int tsii_set_gemm_products(int products) {
    TSII_REQUIRE(products == 0 || products == 3 || products == 6, "set_gemm_products: 0 (f32 MFMA), 3 or 6 (split bf16)");
    tsii::g_products = products;
    return 0;
}

extern "C" int tsii_get_gemm_products(void) { return tsii::g_products; }
