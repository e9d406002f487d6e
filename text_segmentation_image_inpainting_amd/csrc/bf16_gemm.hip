// bf16 ACTIVATION STORAGE, matrix products: the 1x1 convolutions and (as implicit GEMM) the dense k x k convolutions of the
// segmentation nets with bf16 activations / activation gradients in HBM, bf16 x bf16 -> fp32 on v_mfma_f32_32x32x16_bf16
// (models/BaseModels.py:91-127, models/Xception.py:13-114, models/common.py:53-93; BASELINE config 5's "mixed bf16").
//
// Unlike gemm_split.hip there is nothing to split: an operand chunk is 8 bf16 = one 16-byte load that goes to LDS as it is
// (through the producer's BatchNorm + activation when that is fused, K6b), the weights are rounded to bf16 once per call into a
// caller workspace, and the fp32 accumulators are rounded once, on their way out.
//   NT kernels C[M,N] = op(A)[M,K] . B[N,K]^T : forward (A = activations, B = weights) and dX (A = dy, B = weights^T);
//              (plain operands with N, K >= 256 take hgemm_nt_ph_kernel below: 256 x 256 x 64 tiles straight from global memory into LDS)
//              A is a row-major matrix or the im2col VIEW of an NHWC tensor (AMODE 1 forward taps, 2 dX taps).
//              256 threads = 4 waves, 128 x {128, 64, 32} tile, BK = 32, two LDS stages (ONE barrier per K tile), operand image
//              [row][32 bf16] with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3 (conflict free for the staging
//              ds_write_b128 and the fragment ds_read_b128, same image as gemm_split.hip).  Epilogue through LDS: bias, RNE
//              rounding, BatchNorm statistics partials of the ROUNDED values (K6b), the BatchNorm-backward reductions of the
//              consumer-side fusion (K6c), 16-byte stores.
//   TN kernel  C[P,Q] = A[M,P]^T . B[M,Q] (weight gradients): the MFMA wants 8 consecutive m per lane, i.e. both operands
//              transposed: a thread loads a 4(m) x 8(channel) micro-tile of each operand, transposes it in registers (8
//              dword merges per operand and channel pair) and writes 8 bytes per channel -- half of a [channel][8 m] atom at
//              atom(c, ch) = c * 128 + (ch ^ ((ch >> 3) & 7)): a 16-lane ds_write_b64 group (8 channel octets x 2 halves)
//              covers all 32 banks, the fragment ds_read_b128 of a 16-lane group hit 16 distinct atoms.  64-m stages, split-M
//              partial slabs + the shared row reduction.
#include "bf16_common.h"

namespace tsii {

static constexpr int HBK = 32;      // K elements per tile: one LDS row = 32 bf16 = 64 bytes = 4 chunks of 8
__device__ __forceinline__ int h_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// im2col view of an NHWC bf16 tensor [n, h, w, c] (c % 8 == 0): row m = pixel (n, ry, rx) of the row grid [n, rh, rw], column
// k = (tap t = ky * kw + kx, channel ci);  AMODE 1: source (ry*sh - ph + ky*dh, rx*sw - pw + kx*dw);
// AMODE 2 (dX: rows = input pixels, source = dy): ((ry + ph - ky*dh) / sh, (rx + pw - kx*dw) / sw) where divisible.
struct HGather {
    int h, w, c, rh, rw, kw, sh, sw, ph, pw, dh, dw;
};
template <int AMODE>
__device__ __forceinline__ bool h_conv_src(const HGather& cg, int ry, int rx, int ky, int kx, int& sy, int& sx) {
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

struct HEpi {
    const float* bias;      // [N] or NULL
    float* stats;           // [row blocks][4][N]: (count, pivot, sum(y - pivot), sum((y - pivot)^2)) of the STORED values, or NULL
    // K6c (BNB instantiations): the stored values are the gradient w.r.t. a = act(gamma * xhat + beta) of the raw [M, N] bf16
    // tensor bn_y; the epilogue also leaves per row block (sum dz, sum dz * xhat), dz = stored value * act'(z)
    const bf16_t* bn_y;
    const float* bn_mean;
    const float* bn_var;
    const float* bn_gamma;
    const float* bn_beta;
    float bn_eps, bn_neg, bn_hi;
    float* bn_part;         // [row blocks][2][N]
};

template <int ROWS>
struct HChunks { static constexpr int value = (ROWS * 4 + 255) / 256; };

// this thread's chunks of a ROWS x 32 tile of a row-major bf16 matrix: f = tid + 256 i -> row f >> 2, chunk f & 3.
// BRANCH-FREE (clamped addresses, zeroed by the store pass): a branch around a load makes hipcc wait vmcnt(0) per load.
template <int ROWS>
__device__ __forceinline__ void h_load(const bf16_t* __restrict__ Pm, int64_t ld, int64_t row0, int64_t nrows, int k0, int K,
                                       hu32x4 (&regs)[HChunks<ROWS>::value]) {
    const int tid = threadIdx.x;
    const char* __restrict__ base = reinterpret_cast<const char*>(Pm + row0 * ld);
    const int last = (int)((nrows - row0 < ROWS) ? (nrows - row0) : ROWS) - 1;
#pragma unroll
    for (int i = 0; i < HChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        int r = f >> 2;
        r = r < last ? r : last;
        int k = k0 + (f & 3) * 8;
        k = k < K - 8 ? k : K - 8;                                       // K % 8 == 0, K >= 8
        regs[i] = *reinterpret_cast<const hu32x4*>(base + (unsigned)((r * (int)ld + k) * 2));
    }
}

template <int ROWS, bool BNIN>
__device__ __forceinline__ void h_store(unsigned char* __restrict__ S, const hu32x4 (&regs)[HChunks<ROWS>::value], int k0, int nvalid, int K,
                                        const InBN8& bn, float neg, float hi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < HChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const int r = f >> 2, c = f & 3;
        const bool valid = r < nvalid && k0 + c * 8 < K;
        hu32x4 u = regs[i];
        if constexpr (BNIN) {
            float v[8];
            unpack8(u, v);
            apply_inbn8(v, bn, neg, hi);
            u = pack8(v);
        }
        const hu32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<hu32x4*>(S + h_off(r, c)) = valid ? u : z;
    }
}

template <int ROWS>
__device__ __forceinline__ void h_conv_rows(const HGather& cg, int64_t row0, int64_t nrows, int (&rn)[HChunks<ROWS>::value],
                                            int (&ry)[HChunks<ROWS>::value], int (&rx)[HChunks<ROWS>::value]) {
#pragma unroll
    for (int i = 0; i < HChunks<ROWS>::value; ++i) {
        const int64_t row = row0 + ((threadIdx.x + 256 * i) >> 2);
        rn[i] = -1; ry[i] = 0; rx[i] = 0;
        if (row < nrows) {
            rx[i] = (int)(row % cg.rw);
            ry[i] = (int)((row / cg.rw) % cg.rh);
            rn[i] = (int)(row / ((int64_t)cg.rw * cg.rh));
        }
    }
}

// gathered A: the thread's chunk column (8 channels of ONE tap: c % 8 == 0) advances with the K loop, its rows are fixed
template <int ROWS, int AMODE>
__device__ __forceinline__ void h_conv_load(const bf16_t* __restrict__ src, const HGather& cg, const int (&rn)[HChunks<ROWS>::value],
                                            const int (&ry)[HChunks<ROWS>::value], const int (&rx)[HChunks<ROWS>::value], int k0, int K,
                                            hu32x4 (&regs)[HChunks<ROWS>::value], unsigned& okm) {
    int k = k0 + (threadIdx.x & 3) * 8;
    const bool kok = k < K;
    k = kok ? k : K - 8;
    const int t = k / cg.c, ci = k - t * cg.c;
    const int ky = t / cg.kw, kx = t - ky * cg.kw;
    okm = 0u;
#pragma unroll
    for (int i = 0; i < HChunks<ROWS>::value; ++i) {
        int sy = 0, sx = 0;
        const bool ok = kok && rn[i] >= 0 && h_conv_src<AMODE>(cg, ry[i], rx[i], ky, kx, sy, sx);
        const int64_t spix = ok ? ((int64_t)rn[i] * cg.h + sy) * cg.w + sx : 0;
        regs[i] = ld8(src + spix * cg.c + ci);
        okm |= ok ? (1u << i) : 0u;
    }
}
template <int ROWS>
__device__ __forceinline__ void h_conv_store(unsigned char* __restrict__ S, const hu32x4 (&regs)[HChunks<ROWS>::value], unsigned okm) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < HChunks<ROWS>::value; ++i) {
        const int f = tid + 256 * i;
        if (ROWS * 4 % 256 != 0 && f >= ROWS * 4) continue;
        const hu32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<hu32x4*>(S + h_off(f >> 2, f & 3)) = ((okm >> i) & 1u) ? regs[i] : z;
    }
}

// ---- epilogue: accumulators -> LDS band -> (bias, round, statistics / K6c reductions) -> 16-byte bf16 stores --------------
// NTHR threads; the statistics / K6c partial rows describe 128 rows each (tsii_bf16_stat_rows): a taller tile leaves BM / 128 of them.
template <int WM, int WN, int TM, int TN, bool BNB, int SMEM_FLOATS, int NTHR = 256>
__device__ __forceinline__ void h_nt_epilogue(float* __restrict__ smem, hf32x16 (&acc)[TM][TN], bf16_t* __restrict__ C, int64_t ldc,
                                              int64_t M, int N, const HEpi& ep, int64_t m0, int n0, unsigned rowblk) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CS = BN + 4;                       // padded LDS row stride (floats)
    constexpr int BAND = WM * 32;                    // rows staged per pass over t
    constexpr int G = BN / 8;                        // 8-column groups per row
    constexpr int RPP = NTHR / G;                    // rows per store pass
    constexpr int NPASS = BAND / RPP;
    constexpr int HALVES = BM > 128 ? BM / 128 : 1;  // partial rows per tile
    static_assert(BAND * CS <= SMEM_FLOATS, "epilogue band must fit the block's LDS");
    static_assert(BAND % RPP == 0 && NTHR % G == 0, "fixed column group per thread");
    static_assert(2 * RPP * BN + BN <= SMEM_FLOATS, "statistics reduction must fit the block's LDS");
    static_assert(HALVES == 1 || (BM % 128 == 0 && 32 % RPP == 0), "which 128-row half a pass belongs to must not depend on the thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;
    const int g = tid % G, rr0 = tid / G;
    const int col = n0 + g * 8;
    const bool col_ok = col < N;                     // N % 8 == 0: a group is all in or all out
    float bv[8], st1[HALVES][8], st2[HALVES][8], pvt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bv[e] = 0.f; pvt[e] = 0.f;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) { st1[h][e] = 0.f; st2[h][e] = 0.f; }
    }
    if (ep.bias != nullptr && col_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = ep.bias[col + e];
    }
    float bmu[8], bis[8], bga[8], bbe[8];
    if constexpr (BNB) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cc = col_ok ? col + e : 0;
            bmu[e] = ep.bn_mean[cc]; bis[e] = 1.0f / sqrtf(ep.bn_var[cc] + ep.bn_eps);
            bga[e] = ep.bn_gamma[cc]; bbe[e] = ep.bn_beta[cc];
        }
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        int64_t rowv[NPASS];
        hu32x4 yq[BNB ? NPASS : 1];
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int rr = rr0 + RPP * i;
            rowv[i] = m0 + ((rr >> 5) * TM + t) * 32 + (rr & 31);
            if constexpr (BNB) {
                const hu32x4 z = {0u, 0u, 0u, 0u};
                yq[i] = z;
                if (rowv[i] < M && col_ok) yq[i] = ld8(ep.bn_y + rowv[i] * (int64_t)N + col);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CS + (wn * TN + u) * 32 + li] = acc[t][u][r];
        __syncthreads();
        if (t == 0 && ep.stats != nullptr) {
            // block pivot per column: the (rounded) value of the block's first row -- any number typical of the column does, it
            // only has to be the same for every thread of the column group (and is, for every partial row of the tile)
            const float4 q0 = *reinterpret_cast<const float4*>(smem + g * 8), q1 = *reinterpret_cast<const float4*>(smem + g * 8 + 4);
            const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) pvt[e] = bf16_round(qv[e] + bv[e]);
        }
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int rr = rr0 + RPP * i;
            const int hsel = HALVES == 1 ? 0 : ((((RPP * i) >> 5) * TM + t) >> 2);       // rr0 < RPP <= 32: rr >> 5 == (RPP i) >> 5
            const int64_t row = rowv[i];
            if (row >= M || !col_ok) continue;
            const float4 q0 = *reinterpret_cast<const float4*>(smem + rr * CS + g * 8), q1 = *reinterpret_cast<const float4*>(smem + rr * CS + g * 8 + 4);
            float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[e];
            const hu32x4 o = pack8(v);
            if (ep.stats != nullptr || BNB) unpack8(o, v);            // the values as stored
            if (ep.stats != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - pvt[e];
                    st1[hsel][e] += d;
                    st2[hsel][e] = fmaf(d, d, st2[hsel][e]);
                }
            }
            if constexpr (BNB) {
                float ye[8];
                unpack8(yq[i], ye);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (ye[e] - bmu[e]) * bis[e];
                    const float z = fmaf(xh, bga[e], bbe[e]);
                    const float dz = v[e] * inbn_grad(z, ep.bn_neg, ep.bn_hi);
                    st1[hsel][e] += dz;
                    st2[hsel][e] = fmaf(dz, xh, st2[hsel][e]);
                }
            }
            st8_nt(C + row * ldc + col, o);
        }
    }
    if (ep.stats == nullptr && !BNB) return;
    // RPP threads share a column group: combine through LDS, one partial row per 128 rows (per tile when it is shorter).  Layout
    // [sum 0 / 1][e][rr0][g]: the writers' lanes (consecutive g) and the readers' lanes (thread -> (e = tid / G, g = tid % G)) both walk
    // consecutive floats -- as [rr0][g][e][2] a wave's store hit two banks (16-way), 8 us per 256 x 256 tile
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            smem[((0 * 8 + e) * RPP + rr0) * G + g] = st1[h][e];
            smem[((1 * 8 + e) * RPP + rr0) * G + g] = st2[h][e];
        }
        if (rr0 == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) smem[RPP * BN * 2 + e * G + g] = pvt[e];
        }
        __syncthreads();
        const int64_t left = M - (m0 + (HALVES == 1 ? 0 : 128 * h));
        const int re = tid / G, rg = tid % G;                  // this reader's column: 8 rg + re
        const int rcol = n0 + rg * 8 + re;
        if (tid < BN && rcol < N && left > 0) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
            for (int j = 0; j < RPP; ++j) { a1 += smem[((0 * 8 + re) * RPP + j) * G + rg]; a2 += smem[((1 * 8 + re) * RPP + j) * G + rg]; }
            const int64_t prow = (int64_t)rowblk * HALVES + h;
            if constexpr (BNB) {
                float* sp = ep.bn_part + prow * 2 * N;
                sp[rcol] = a1;
                sp[N + rcol] = a2;
            } else {
                float* sp = ep.stats + prow * 4 * N;
                constexpr int HR = HALVES == 1 ? BM : 128;
                sp[rcol] = (float)(left < HR ? left : HR);
                sp[N + rcol] = smem[RPP * BN * 2 + re * G + rg];
                sp[2 * N + rcol] = a1;
                sp[3 * N + rcol] = a2;
            }
        }
    }
}

static constexpr int HNT_LBN_MAXK = 1024;
template <int WM, int WN, int TM, int TN, int AMODE, bool BNIN, bool BNB>
#ifndef HNT_WAVES
#define HNT_WAVES 4      // waves per SIMD the NT kernel is held to (A/B: tools/variants; 1 = whatever the compiler takes)
#endif
__global__ __launch_bounds__(256, TM * TN >= 8 ? 2 : ((HNT_WAVES > 3 && BNB) ? 3 : HNT_WAVES)) void hgemm_nt_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                       bf16_t* __restrict__ C, int64_t ldc, int64_t M, int N, int K, HEpi ep, unsigned ntn,
                                                       InBN ib, HGather cg) {
    constexpr bool CONV = AMODE != 0;
    static_assert(!CONV || !BNIN, "the gathered operand has no BatchNorm-on-load form");
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STAGE_BYTES = (BM + BN) * 64;
    constexpr int OP_FLOATS = 2 * STAGE_BYTES / 4;               // two stages
    constexpr int EP_FLOATS = WM * 32 * (BN + 4);
    constexpr int ST_FLOATS = 2 * (256 / (BN / 8)) * BN + BN;
    constexpr int SMEM_FLOATS = OP_FLOATS > EP_FLOATS ? (OP_FLOATS > ST_FLOATS ? OP_FLOATS : ST_FLOATS) : (EP_FLOATS > ST_FLOATS ? EP_FLOATS : ST_FLOATS);
    // the 128 x 256 form keeps the BatchNorm-on-load constants of all K input channels in LDS (K <= HNT_LBN_MAXK, checked at the launch):
    // fetched per K tile from global memory they were four 16-byte loads per thread beside the two of its operand chunks
    constexpr bool LBN = BNIN && TM * TN >= 8;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS + (LBN ? 2 * HNT_LBN_MAXK : 0)];
    unsigned char* S0 = reinterpret_cast<unsigned char*>(smem);
    float* lbn = smem + SMEM_FLOATS;
    auto fetch_bn = [&](int k, InBN8& o) {
        if constexpr (LBN) {
            const float4 a0 = *reinterpret_cast<const float4*>(lbn + k), a1 = *reinterpret_cast<const float4*>(lbn + k + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(lbn + HNT_LBN_MAXK + k), b1 = *reinterpret_cast<const float4*>(lbn + HNT_LBN_MAXK + k + 4);
            o.sc[0] = a0.x; o.sc[1] = a0.y; o.sc[2] = a0.z; o.sc[3] = a0.w; o.sc[4] = a1.x; o.sc[5] = a1.y; o.sc[6] = a1.z; o.sc[7] = a1.w;
            o.sh[0] = b0.x; o.sh[1] = b0.y; o.sh[2] = b0.z; o.sh[3] = b0.w; o.sh[4] = b1.x; o.sh[5] = b1.y; o.sh[6] = b1.z; o.sh[7] = b1.w;
        } else {
            load_inbn8(ib, k, o);
        }
    };
    if constexpr (LBN) {
        for (int i = threadIdx.x; i < K; i += 256) { lbn[i] = ib.sc[i]; lbn[HNT_LBN_MAXK + i] = ib.sh[i]; }
        __syncthreads();
    }

    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(bid / ntn) * BM;
    const int n0 = (int)(bid % ntn) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    hf32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    constexpr int NA = HChunks<BM>::value, NB = HChunks<BN>::value;
    const int mvalid = (int)((M - m0 < BM) ? (M - m0) : BM), nvalid = (N - n0 < BN) ? (N - n0) : BN;
    hu32x4 ra[NA], rb[NB];
    int rn[NA], ry[NA], rx[NA];
    unsigned okm = 0u;
    InBN8 bn;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bn.sc[e] = 0.f; bn.sh[e] = 0.f; }
    const int pk = (tid & 3) * 8;
    if constexpr (CONV) {
        h_conv_rows<BM>(cg, m0, M, rn, ry, rx);
        h_conv_load<BM, AMODE>(A, cg, rn, ry, rx, 0, K, ra, okm);
    } else {
        h_load<BM>(A, lda, m0, M, 0, K, ra);
    }
    h_load<BN>(B, K, n0, N, 0, K, rb);
    if constexpr (BNIN) fetch_bn(pk < K ? pk : K - 8, bn);
    if constexpr (CONV) h_conv_store<BM>(S0, ra, okm);
    else h_store<BM, BNIN>(S0, ra, 0, mvalid, K, bn, ib.neg, ib.hi);
    h_store<BN, false>(S0 + BM * 64, rb, 0, nvalid, K, bn, 1.f, 0.f);
    __syncthreads();

    // fragment addresses: row li of a 32-row tile, k-chunk 2s + hi; the swizzle term depends on li only
    const int swz = (li >> 2) & 3;
    const int fo0 = li * 64 + (((0 + hi) ^ swz) << 4), fo1 = li * 64 + (((2 + hi) ^ swz) << 4);

    const int nk = (K + HBK - 1) / HBK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        unsigned char* Sc = S0 + (kt & 1) * STAGE_BYTES;
        unsigned char* Sn = S0 + ((kt + 1) & 1) * STAGE_BYTES;
        if (more) {          // the next tile's global loads fly during the MFMA phase
            if constexpr (CONV) h_conv_load<BM, AMODE>(A, cg, rn, ry, rx, (kt + 1) * HBK, K, ra, okm);
            else h_load<BM>(A, lda, m0, M, (kt + 1) * HBK, K, ra);
            h_load<BN>(B, K, n0, N, (kt + 1) * HBK, K, rb);
        }
        const unsigned char* Aw = Sc + (wm * TM) * 32 * 64;
        const unsigned char* Bw = Sc + BM * 64 + (wn * TN) * 32 * 64;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int fo = s == 0 ? fo0 : fo1;
            hbf16x8 b[TN];
#pragma unroll
            for (int u = 0; u < TN; ++u) b[u] = __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(Bw + u * (32 * 64) + fo));
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const hbf16x8 a = __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(Aw + t * (32 * 64) + fo));
#pragma unroll
                for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[u], acc[t][u], 0, 0, 0);
            }
        }
        if (more) {
            if constexpr (BNIN) {       // (scale, shift) of the next tile's channels: L2 hits, fetched behind the MFMA phase
                const int k = (kt + 1) * HBK + pk;
                fetch_bn(k < K ? k : K - 8, bn);
            }
            // the other stage was last read one iteration ago, before the barrier every wave has passed since
            if constexpr (CONV) h_conv_store<BM>(Sn, ra, okm);
            else h_store<BM, BNIN>(Sn, ra, (kt + 1) * HBK, mvalid, K, bn, ib.neg, ib.hi);
            h_store<BN, false>(Sn + BM * 64, rb, (kt + 1) * HBK, nvalid, K, bn, 1.f, 0.f);
        }
        __syncthreads();
    }
    h_nt_epilogue<WM, WN, TM, TN, BNB, SMEM_FLOATS>(smem, acc, C, ldc, M, N, ep, m0, n0, bid / ntn);
}

__device__ __attribute__((aligned(16))) const unsigned h_zero_page[4] = {0u, 0u, 0u, 0u};

// ---- NT, 256 x 256 x 64 tiles, two wave groups in counter-phase, operands staged by row halves ------------------------------------
// For the matrix-core-bound products (N, K >= 256: the middle / exit flow's 512 x 512 layers) the 128-row kernel spends as long moving
// an operand tile through registers into LDS (13 cycles per ds_write_b128) and reading it back (one KB per MFMA with 64 x 64 wave
// tiles) as the matrix pipe spends on it, and its four waves run in lock step: 0.68 PF/s on 65536 x 4096 x 4096.  Here: 256 x 256 x 64
// tiles, 512 threads = 8 waves with 128 x 64 wave tiles (0.75 KB of fragment reads per MFMA); operands by global_load_lds_dwordx4
// -- no staging registers, no ds_write: a wave instruction fills 8 rows x 128 bytes, lane l fetching for LDS position (row l >> 3,
// slot l & 7) the k-chunk (l & 7) ^ ((row >> 1) & 7): the XOR swizzle lives on the SOURCE address, the LDS side is lane-linear, and the
// fragment ds_read_b128 (a lane group = 16 rows of one chunk column) covers all 64 banks; K tails come from a page of zeros.
// Two 64 KB K tiles fit, and a stage cannot be refilled as a whole without the load waiting for itself.  Following the 8-phase
// schedule of the CDNA4 guide (§ "The 256² 8-phase template"), a K tile is
// four half-tiles of 128 rows (A0, A1, B0, B1: 16 KB = two load instructions per wave each) and a wave's work on it four PHASES, one
// 64 x 32 quadrant of its 128 x 64 tile each (8 MFMAs over the whole K tile):
//     phase   L section: fragment reads of K tile kt    half-tile staged (two loads    C section: 8 MFMAs
//                                                      between the MFMAs of C)
//     P1      a0 (8 reads), b0 (4)            A0 of K tile kt + 1                   (a0, b0)
//     P2      b1 (4)                          A1 of kt + 1                          (a0, b1)
//     P3      a1 (8, over a0)                 B0 of kt + 2                          (a1, b1)
//     P4      --     vmcnt: all of kt + 1     B1 of kt + 2                          (a1, b0)    b0 is kept, not read again
// A half-tile is refilled as soon as its last reader is done: the B halves of a buffer are free after P2, the A halves after P3.
// Group 1 (waves 4 .. 7, the SIMD partners of waves 0 .. 3) runs one barrier interval behind group 0, so a SIMD always has one wave
// in a C section.  RAW: the youngest half-tile of K tile kt + 1 (A1) is issued in C(P2) and waited for at the end of L(P4) -- three
// intervals later, with B0 of kt + 2 still in flight (never a vmcnt(0) in the loop) -- which for group 1 is the interval before group 0
// reads it;
// WAR: every L section retires its reads (lgkmcnt(0)) in front of its barrier, and a half-tile is restaged at the earliest two
// intervals after group 0's / one after group 1's last read.
// Measured (tools/bf16_bench.py --gemm, profiles/r05r_gemm_fill.log): 65536 x 4096 x 4096 1.17 PF/s with random operands, 1.38 with
// zeros (the part is power-capped: the sustained clock depends on how many operand bits toggle), 8192^3 1.06 / 1.10.  Two other
// structures were built and measured first (profiles/r05m_nt_dl.log, r05n_nt_forms.log, r05p_nt_ph.log; commit 85e7021 has their
// source): 8 waves in lock step, two whole-tile stages (0.91 - 0.98 PF/s: behind every barrier all waves fetch with nothing in the
// matrix pipe), and the counter-phase scheme on K slabs of 32 in a 4-stage ring (1.0 - 1.05, but 0.74 at K = 8192: 64 bytes of a 16 KB
// row per request).  BatchNorm-on-load at fragment time is VALU-bound on all of them (218 us against 136 on 131072 x 512 x 512).
// AMODE 1 / 2: A is the im2col view of an NHWC tensor (forward / dX taps of a dense convolution, HGather): the LDS-DMA takes a
// per-lane source address anyway, so the gather is the address -- a K tile of 64 channels lies inside ONE tap (c % 64 == 0, checked at
// the launch), i.e. tap and channel offset are block-uniform per K tile and a lane only keeps the pixel of its four A rows; taps
// outside the image fetch the page of zeros.
template <int AMODE, bool BNB>
__global__ __launch_bounds__(512, 1) void hgemm_nt_ph_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                             bf16_t* __restrict__ C, int64_t ldc, int64_t M, int N, int K, HEpi ep, unsigned ntn, HGather cg) {
    constexpr bool CONV = AMODE != 0;
    constexpr int WM = 2, WN = 4, TM = 4, TN = 2;
    constexpr int BM = 256, BN = 256, BK = 64, RB = 128;
    constexpr int HALF_BYTES = 128 * RB, BUF_BYTES = 4 * HALF_BYTES;                        // 16 KB, 64 KB
    constexpr int OP_FLOATS = 2 * BUF_BYTES / 4;
    constexpr int EP_FLOATS = WM * 32 * (BN + 4);
    constexpr int ST_FLOATS = 2 * (512 / (BN / 8)) * BN + BN;
    constexpr int SMEM_FLOATS = OP_FLOATS > EP_FLOATS ? (OP_FLOATS > ST_FLOATS ? OP_FLOATS : ST_FLOATS) : (EP_FLOATS > ST_FLOATS ? EP_FLOATS : ST_FLOATS);
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    unsigned char* S0 = reinterpret_cast<unsigned char*>(smem);

    // tile order inside an XCD's share of the grid: groups of PH_GROUP_M row tiles x all column tiles, walked row-tile-fastest, so that
    // the ~32 blocks an XCD runs at a time cover a squarish patch (8 x 4 tiles: 12 operand panels per K tile through its L2 instead of
    // the 2 x 16 = 18 of the row-major order)
#ifndef PH_GROUP_M
#define PH_GROUP_M 8
#endif
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    unsigned tm, tn;
    if (PH_GROUP_M > 1) {
        const unsigned ntm = gridDim.x / ntn;
        const unsigned per = PH_GROUP_M * ntn;
        const unsigned gid = bid / per, first = gid * PH_GROUP_M;
        const unsigned gm = ntm - first < (unsigned)PH_GROUP_M ? ntm - first : (unsigned)PH_GROUP_M;
        tm = first + (bid % per) % gm;
        tn = (bid % per) / gm;
    } else {
        tm = bid / ntn; tn = bid % ntn;
    }
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, hi = lane >> 5;

    hf32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // this lane's two pieces of each half-tile (index 2 h + j; h: A0, A1, B0, B1): instruction j of wave w fills rows 8 (8 j + w) .. + 7
    const bf16_t* gsrc[8];
    int apn[4], apyx[4];                                         // gathered A: image and (y << 16 | x) of this lane's four A rows, -1: past M
    const int kc = ((lane & 7) ^ ((((wave & 1) << 3) + (lane >> 3)) >> 1 & 7)) * 8;         // slot ^ ((row >> 1) & 7), row = 8 (8 j + w) + (lane >> 3)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (h & 1) * 128 + 8 * (8 * j + wave) + (lane >> 3);                 // row of the 256-row operand tile
            if (h < 2) {         // rows past the matrix re-read the last one: their products are never stored
                const int64_t row = m0 + r;
                if constexpr (CONV) {
                    gsrc[2 * h + j] = A + kc;
                    apn[2 * h + j] = -1; apyx[2 * h + j] = 0;
                    if (row < M) {
                        const int64_t q = row / cg.rw;
                        apn[2 * h + j] = (int)(q / cg.rh);
                        apyx[2 * h + j] = ((int)(q % cg.rh) << 16) | (int)(row - q * cg.rw);
                    }
                } else {
                    gsrc[2 * h + j] = A + (row < M ? row : M - 1) * lda + kc;
                }
            } else {
                const int row = n0 + r;
                gsrc[2 * h + j] = B + (int64_t)(row < N ? row : N - 1) * K + kc;
            }
        }
    const bf16_t* zpage = reinterpret_cast<const bf16_t*>(h_zero_page);
    auto stage_piece = [&](int kt, int h, int j) {   // BRANCH-FREE: a K tile past the end fetches zeros into a half nobody reads again
        unsigned char* S = S0 + (kt & 1) * BUF_BYTES + h * HALF_BYTES + wave * 1024 + j * 8192;
        const int k0 = kt * BK;
        if (CONV && h < 2) {
            const int kk = k0 < K ? k0 : 0;                      // block-uniform: the K tile's tap and first channel
            const int t = kk / cg.c, ci0 = kk - t * cg.c;
            const int ky = t / cg.kw, kx = t - ky * cg.kw;
            int sy = 0, sx = 0;
            const bool ok = k0 < K && apn[2 * h + j] >= 0 && h_conv_src<AMODE == 2 ? 2 : 1>(cg, apyx[2 * h + j] >> 16, apyx[2 * h + j] & 0xffff, ky, kx, sy, sx);
            async_load16_lds(S, ok ? gsrc[2 * h + j] + (((int64_t)apn[2 * h + j] * cg.h + sy) * cg.w + sx) * cg.c + ci0 : zpage);
        } else {
            async_load16_lds(S, (k0 + kc < K) ? gsrc[2 * h + j] + k0 : zpage);
        }
    };
    auto stage_half = [&](int kt, int h) { stage_piece(kt, h, 0); stage_piece(kt, h, 1); };
#ifndef PH_GLDS_IN_C
#define PH_GLDS_IN_C 1   // the half-tile's two load instructions between the MFMAs of the C section (1) or in the L section (0)
#endif
    // 8 MFMAs of one quadrant; with PH_GLDS_IN_C the two pieces of half-tile (skt, sh) go behind the 2nd and the 6th (an LDS-DMA issue
    // costs its wave 60 .. 180 cycles: in the L section that is time the partner's MFMAs cannot hide, here only its own pipe slack)
#ifndef PH_PRIO
#define PH_PRIO 1
#endif
#define TSII_PH_QUADRANT(AT, BU, FB, SKT, SH)                                                                                          \
        __builtin_amdgcn_s_setprio(PH_PRIO);                                                                                                  \
        _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                                \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                             \
                acc[AT + t][BU] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, fra[t][s2]),                      \
                                                                          __builtin_bit_cast(hbf16x8, FB[s2]), acc[AT + t][BU], 0, 0, 0); \
                if (PH_GLDS_IN_C && s2 == 0 && t == 1) { __builtin_amdgcn_sched_barrier(0); stage_piece(SKT, SH, 0); __builtin_amdgcn_sched_barrier(0); } \
                if (PH_GLDS_IN_C && s2 == 2 && t == 1) { __builtin_amdgcn_sched_barrier(0); stage_piece(SKT, SH, 1); __builtin_amdgcn_sched_barrier(0); } \
            }                                                                                                                           \
        __builtin_amdgcn_s_setprio(0);
    const int nkt = (K + BK - 1) / BK;
    stage_half(0, 0); stage_half(0, 1); stage_half(0, 2); stage_half(0, 3);
    stage_half(1, 2); stage_half(1, 3);

    // fragment addresses inside a buffer: row li of a 32-row tile, k-chunk 2 s + hi at slot (2 s + hi) ^ ((li >> 1) & 7)
    const int swz = (li >> 1) & 7;
    const int fa = wm * HALF_BYTES + li * RB;                                              // this wave's rows are exactly A half wm
    const int fb = (2 + (wn >> 1)) * HALF_BYTES + ((wn & 1) * 64 + li) * RB;
    int co[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) co[s2] = ((2 * s2 + hi) ^ swz) << 4;

    async_wait_lds<4>();                          // this wave's pieces of K tile 0 (the B halves of K tile 1 may be in flight)
    lds_barrier();
    if (grp == 1) lds_barrier();                  // group 1 runs one interval behind

    hu32x4 fra[2][4], frb0[4], frb1[4];
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* Sa = S0 + (kt & 1) * BUF_BYTES + fa;
        const unsigned char* Sb = S0 + (kt & 1) * BUF_BYTES + fb;
        // ---- P1 ---------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) lds_read16<0>(frb0[s2], Sb + co[s2]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { lds_read16<0>(fra[0][s2], Sa + co[s2]); lds_read16<32 * RB>(fra[1][s2], Sa + co[s2]); }
        if (!PH_GLDS_IN_C) stage_half(kt + 1, 0);
        lds_wait<0>(fra[0][0], fra[0][1], fra[0][2], fra[0][3], frb0[0], frb0[1]);
        lds_wait<0>(fra[1][0], fra[1][1], fra[1][2], fra[1][3], frb0[2], frb0[3]);
        lds_barrier();
        TSII_PH_QUADRANT(0, 0, frb0, kt + 1, 0)
        lds_barrier();
        // ---- P2 ---------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) lds_read16<32 * RB>(frb1[s2], Sb + co[s2]);
        if (!PH_GLDS_IN_C) stage_half(kt + 1, 1);
        lds_wait<0>(frb1[0], frb1[1], frb1[2], frb1[3], fra[0][0], fra[1][0]);
        lds_barrier();
        TSII_PH_QUADRANT(0, 1, frb1, kt + 1, 1)
        lds_barrier();
        // ---- P3 ---------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { lds_read16<64 * RB>(fra[0][s2], Sa + co[s2]); lds_read16<96 * RB>(fra[1][s2], Sa + co[s2]); }
        if (!PH_GLDS_IN_C) stage_half(kt + 2, 2);
        lds_wait<0>(fra[0][0], fra[0][1], fra[0][2], fra[0][3], frb1[0], frb1[1]);
        lds_wait<0>(fra[1][0], fra[1][1], fra[1][2], fra[1][3], frb1[2], frb1[3]);
        lds_barrier();
        TSII_PH_QUADRANT(2, 1, frb1, kt + 2, 2)
        lds_barrier();
        // ---- P4 ---------------------------------------------------------------------------------------------------------------
        if (!PH_GLDS_IN_C) stage_half(kt + 2, 3);
        // everything up to A1 of K tile kt + 1 has landed; still in flight: B0 of kt + 2 (and B1 when it was issued above)
        async_wait_lds<PH_GLDS_IN_C ? 2 : 4>();
        lds_barrier();
        TSII_PH_QUADRANT(2, 0, frb0, kt + 2, 3)
        lds_barrier();
    }
#undef TSII_PH_QUADRANT
    if (grp == 0) lds_barrier();
    async_wait_lds<0>();                          // the zero-page pieces of the K tiles past the end: nothing may land in the epilogue's LDS
    h_nt_epilogue<WM, WN, TM, TN, BNB, SMEM_FLOATS, 512>(smem, acc, C, ldc, M, N, ep, m0, n0, tm);
}

// ---- TN (weight gradients) ------------------------------------------------------------------------------------------------
static constexpr int HTN_STAGE = 64;         // m rows per stage
__device__ __forceinline__ int htn_atom(int c, int ch) { return c * 128 + (ch ^ ((ch >> 3) & 7)); }

// this thread's 4(m) x 8(channel) micro-tile of a row-major bf16 matrix; rows / columns outside are clamped (zeroed at the store)
__device__ __forceinline__ void htn_load(const bf16_t* __restrict__ Pm, int64_t ld, int64_t m0, int64_t mend, int c0, int ncols, int co, int mq,
                                         hu32x4 (&regs)[4]) {
    const char* __restrict__ base = reinterpret_cast<const char*>(Pm + m0 * ld);
    const int last = (int)((mend - m0 < HTN_STAGE) ? (mend - m0) : HTN_STAGE) - 1;
    int c = c0 + co * 8;
    c = c < ncols - 8 ? c : ncols - 8;                                   // ncols % 8 == 0, ncols >= 8
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        int r = mq * 4 + jj;
        r = r < last ? r : last;
        regs[jj] = *reinterpret_cast<const hu32x4*>(base + ((int64_t)r * ld + c) * 2);
    }
}
// gathered B (dW of a dense conv): the thread's column octet is ONE (tap, 8 channels) for the whole block, rows = output pixels
__device__ __forceinline__ void htn_conv_load(const bf16_t* __restrict__ src, const HGather& cg, int64_t m0, int64_t mend, int ky, int kx, int ci,
                                              bool kok, int mq, hu32x4 (&regs)[4], unsigned& okm) {
    okm = 0u;
    const int64_t last = mend - 1;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        int64_t row = m0 + mq * 4 + jj;
        const bool in = row <= last;
        row = in ? row : last;
        const int64_t q = row / cg.rw;
        const int rx = (int)(row - q * cg.rw);
        const int64_t n = q / cg.rh;
        const int ry = (int)(q - n * cg.rh);
        int sy = 0, sx = 0;
        const bool ok = in && kok && h_conv_src<1>(cg, ry, rx, ky, kx, sy, sx);
        const int64_t spix = ok ? (n * cg.h + sy) * cg.w + sx : 0;
        regs[jj] = ld8(src + spix * cg.c + ci);
        okm |= ok ? (1u << jj) : 0u;
    }
}
// (BatchNorm + activation) -> zero outside -> 4 x 8 register transpose -> 8 bytes per channel into the [channel][8 m] atoms
template <bool BNIN>
__device__ __forceinline__ void htn_store(unsigned char* __restrict__ S, hu32x4 (&regs)[4], int co, int mq, unsigned rowmask, bool col_ok,
                                          const InBN8& bn, float neg, float hi) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if constexpr (BNIN) {
            float v[8];
            unpack8(regs[jj], v);
            apply_inbn8(v, bn, neg, hi);
            regs[jj] = pack8(v);
        }
        const hu32x4 z = {0u, 0u, 0u, 0u};
        regs[jj] = (col_ok && ((rowmask >> jj) & 1u)) ? regs[jj] : z;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = e >> 1;
        unsigned d0, d1;
        if ((e & 1) == 0) {
            d0 = (regs[0][j] & 0xffffu) | (regs[1][j] << 16);
            d1 = (regs[2][j] & 0xffffu) | (regs[3][j] << 16);
        } else {
            d0 = (regs[0][j] >> 16) | (regs[1][j] & 0xffff0000u);
            d1 = (regs[2][j] >> 16) | (regs[3][j] & 0xffff0000u);
        }
        const int off = htn_atom(mq >> 1, co * 8 + e) * 16 + (mq & 1) * 8;
        *reinterpret_cast<hu32x2*>(S + off) = hu32x2{d0, d1};
    }
}

template <bool BNIN, bool BCONV>
__global__ __launch_bounds__(256) void hgemm_tn_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ Cws, int64_t M, int Pn, int Q, int64_t chunk, InBN ib,
                                                       unsigned qtiles, unsigned ptiles, HGather cg) {
    static_assert(!BCONV || !BNIN, "the gathered operand has no BatchNorm-on-load form");
    constexpr int OPB = 8 * 128 * 16;                  // one operand stage: 8 m-chunks x 128 channels x 16 bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OPB];
    unsigned char* As = smem;
    unsigned char* Bs = smem + OPB;

    // 1-D grid, XCD-aware: the (p, q) tiles of one m-chunk get consecutive logical ids = one XCD (their panel re-reads hit its L2)
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned zsplit = bid / (qtiles * ptiles), rem = bid % (qtiles * ptiles);
    const int q0 = (int)(rem % qtiles) * 128, p0 = (int)(rem / qtiles) * 128;
    const int64_t mbeg = (int64_t)zsplit * chunk;
    const int64_t mend = (mbeg + chunk < M) ? mbeg + chunk : M;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;
    const int co = (lane & 7) | (((lane >> 4) & 1) << 3);                       // channel octet 0..15
    const int mq = ((lane >> 3) & 1) | ((lane >> 5) << 1) | (wave << 2);        // m quad 0..15

    hf32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    InBN8 bn;
#pragma unroll
    for (int e = 0; e < 8; ++e) { bn.sc[e] = 0.f; bn.sh[e] = 0.f; }
    const bool acol_ok = p0 + co * 8 < Pn, bcol_ok = q0 + co * 8 < Q;
    if constexpr (BNIN) load_inbn8(ib, bcol_ok ? q0 + co * 8 : Q - 8, bn);     // this thread's 8 B columns, fixed for the block
    int cky = 0, ckx = 0, cci = 0;
    if constexpr (BCONV) {
        const int k = bcol_ok ? q0 + co * 8 : Q - 8;
        const int t = k / cg.c;
        cci = k - t * cg.c;
        cky = t / cg.kw; ckx = t - cky * cg.kw;
    }
    hu32x4 ra[4], rb[4];
    unsigned okm = 0u;
    auto rowmask = [&](int64_t m0s) -> unsigned {
        const int left = (int)((mend - m0s < HTN_STAGE) ? (mend - m0s) : HTN_STAGE);
        unsigned mk = 0u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) mk |= (mq * 4 + jj < left) ? (1u << jj) : 0u;
        return mk;
    };
    auto load_stage = [&](int64_t m0s) {
        htn_load(A, lda, m0s, mend, p0, Pn, co, mq, ra);
        if constexpr (BCONV) htn_conv_load(B, cg, m0s, mend, cky, ckx, cci, bcol_ok, mq, rb, okm);
        else htn_load(B, ldb, m0s, mend, q0, Q, co, mq, rb);
    };
    auto store_stage = [&](int64_t m0s) {
        const unsigned mk = rowmask(m0s);
        htn_store<false>(As, ra, co, mq, mk, acol_ok, bn, 1.f, 0.f);
        if constexpr (BCONV) htn_store<false>(Bs, rb, co, mq, okm, bcol_ok, bn, 1.f, 0.f);
        else htn_store<BNIN>(Bs, rb, co, mq, mk, bcol_ok, bn, ib.neg, ib.hi);
    };
    load_stage(mbeg);
    store_stage(mbeg);
    __syncthreads();

    int foA[2][4], foB[2][4];          // fragment byte offsets: channel li of wave tile t, m-chunk 2s + hi
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            foA[t][s] = htn_atom(2 * s + hi, (wm * 2 + t) * 32 + li) * 16;
            foB[t][s] = htn_atom(2 * s + hi, (wn * 2 + t) * 32 + li) * 16;
        }
    auto mfma_phase = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            hbf16x8 b[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(Bs + foB[u][s]));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const hbf16x8 a = __builtin_bit_cast(hbf16x8, *reinterpret_cast<const hu32x4*>(As + foA[t][s]));
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[u], acc[t][u], 0, 0, 0);
            }
        }
    };
    // every iteration loads the NEXT stage unconditionally (the last stage is peeled)
    for (int64_t mt = mbeg; mt + HTN_STAGE < mend; mt += HTN_STAGE) {
        load_stage(mt + HTN_STAGE);
        mfma_phase();
        __syncthreads();
        store_stage(mt + HTN_STAGE);
        __syncthreads();
    }
    mfma_phase();

    float* Cz = Cws + (int64_t)zsplit * Pn * Q;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + (wm * 2 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (p >= Pn) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + (wn * 2 + u) * 32 + li;
                if (q < Q) Cz[(int64_t)p * Q + q] = acc[t][u][r];
            }
        }
}

// ---- TN, 256 x 256 tiles: operands straight into LDS in their memory layout, fragments by the transposing LDS read -----------------
// hgemm_tn_kernel above moves both operands through a register transpose and ds_write_b64 into [channel][8 m] atoms -- the LDS pipe,
// not the matrix pipe, sets its time (0.34 - 0.45 PF/s).  gfx950's ds_read_b64_tr_b16 removes the transpose: the operands are
// staged AS THEY LIE IN MEMORY ([m][channel] rows, global_load_lds_dwordx4, no registers, no ds_write) and a lane reads four consecutive
// m of its channel with one instruction (two per MFMA operand).  Everything else is hgemm_nt_ph_kernel: 256 x 256 tiles (P x Q
// channels), K tiles of 64 m as four half-tiles (A0, A1: channels p0 + 0 / 128 ..; B0, B1), a half-tile = [64 m][128 channels] = 64 rows
// of 256 bytes with the 64-byte segment c of row r at slot c ^ (r & 3) (a 32-lane half of the transposing read covers 4 rows x 64
// bytes: four different slots = all 64 banks), four quadrant phases, two wave groups in counter-phase, the same RAW / WAR argument.
// BatchNorm-on-load (the B operand is x behind BatchNorm + activation) is applied to the B FRAGMENT: a lane's channel is fixed
// (column l & 31 of its two B tiles), so scale / shift are four registers, no table.  The gathered B operand of a dense convolution
// (im2col view, one tap per 128-channel half: cin % 128 == 0) differs in the source address only; taps outside the image and rows past
// the chunk come from the page of zeros.  Split-M partial slabs in fp32 like the kernel above.
template <bool BNIN, bool BCONV>
__global__ __launch_bounds__(512, 1) void hgemm_tn_ph_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ Cws, int64_t M, int Pn, int Q, int64_t chunk, InBN ib,
                                                             unsigned qtiles, unsigned ptiles, HGather cg) {
    static_assert(!BCONV || !BNIN, "the gathered operand has no BatchNorm-on-load form");
    constexpr int HALF_BYTES = 64 * 256, BUF_BYTES = 4 * HALF_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char S0[2 * BUF_BYTES];

    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned zsplit = bid / (qtiles * ptiles), rem = bid % (qtiles * ptiles);
    const int q0 = (int)(rem % qtiles) * 256, p0 = (int)(rem / qtiles) * 256;
    const int64_t mbeg = (int64_t)zsplit * chunk;
    const int64_t mend = (mbeg + chunk < M) ? mbeg + chunk : M;
    const int nkt = (int)((mend - mbeg + 63) / 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 2, wn = wave & 3;                     // 2 (P) x 4 (Q) waves, wave tile 128 x 64
    const int li = lane & 31, hi = lane >> 5;

    hf32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // ---- staging: piece j (0, 1) of wave w fills rows 4 (8 j + w) .. + 3 of a half-tile; lane: row + (lane >> 4), 16-byte slot lane & 15
    const int sr = lane >> 4;                                    // = row & 3
    const int chunk16 = ((((lane & 15) >> 2) ^ sr) << 2) | (lane & 3);      // the logical 8-channel chunk this slot holds
    const int rj0 = 4 * wave + sr, rj1 = 4 * (8 + wave) + sr;    // this lane's rows of a K tile
    const int colA0 = p0 + chunk16 * 8, colB0 = q0 + chunk16 * 8;            // channel of the chunk in half 0 (half 1: + 128)
    int bky[2] = {0, 0}, bkx[2] = {0, 0}, bci[2] = {0, 0};
    if constexpr (BCONV) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = colB0 + 128 * h < Q ? colB0 + 128 * h : 0;
            const int t = k / cg.c;
            bci[h] = k - t * cg.c;
            bky[h] = t / cg.kw; bkx[h] = t - bky[h] * cg.kw;
        }
    }
    const bf16_t* zpage = reinterpret_cast<const bf16_t*>(h_zero_page);
    // BRANCH-FREE: rows past the chunk, channels past the matrix, taps outside the image and K tiles past the end fetch zeros
    auto stage_piece = [&](int kt, int h, int j) {
        unsigned char* S = S0 + (kt & 1) * BUF_BYTES + h * HALF_BYTES + (8 * j + wave) * 1024;
        const int64_t m = mbeg + (int64_t)kt * 64 + (j == 0 ? rj0 : rj1);
        const bool mok = m < mend;
        const bf16_t* src;
        if (h < 2) {
            const int col = colA0 + 128 * h;
            src = (mok && col < Pn) ? A + m * lda + col : zpage;
        } else if constexpr (BCONV) {
            const int hh = h - 2;
            const int64_t mm = mok ? m : 0;
            const int64_t qq = mm / cg.rw;
            const int rx = (int)(mm - qq * cg.rw);
            const int64_t n = qq / cg.rh;
            const int ry = (int)(qq - n * cg.rh);
            int sy = 0, sx = 0;
            const bool ok = mok && colB0 + 128 * hh < Q && h_conv_src<1>(cg, ry, rx, bky[hh], bkx[hh], sy, sx);
            src = ok ? B + ((n * cg.h + sy) * cg.w + sx) * cg.c + bci[hh] : zpage;
        } else {
            const int col = colB0 + 128 * (h - 2);
            src = (mok && col < Q) ? B + m * ldb + col : zpage;
        }
        async_load16_lds(S, src);
    };
    auto stage_half = [&](int kt, int h) { stage_piece(kt, h, 0); stage_piece(kt, h, 1); };
    stage_half(0, 0); stage_half(0, 1); stage_half(0, 2); stage_half(0, 3);
    stage_half(1, 2); stage_half(1, 3);

    // ---- fragments: tile of 32 channels = 64-byte segment c of the half-tile's rows; lane (q = lane & 15, g1 = (lane >> 4) & 1, hi) points
    // at row 16 s + 8 hi + (q >> 2) [+ 4 for the second read], bytes 32 g1 + 8 (q & 3) of segment c ^ (q >> 2)
    const int q4 = (lane & 15) >> 2;
    const int fbase = (8 * hi + q4) * 256 + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    int offA[4], offB[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) offA[t] = wm * HALF_BYTES + fbase + ((t ^ q4) << 6);
#pragma unroll
    for (int u = 0; u < 2; ++u) offB[u] = (2 + (wn >> 1)) * HALF_BYTES + fbase + (((2 * (wn & 1) + u) ^ q4) << 6);
    float bsc[2] = {0.f, 0.f}, bsh[2] = {0.f, 0.f};
    if constexpr (BNIN) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = q0 + wn * 64 + u * 32 + li;
            bsc[u] = ib.sc[c < Q ? c : Q - 1]; bsh[u] = ib.sh[c < Q ? c : Q - 1];
        }
    }
    const float bneg = ib.neg, bhi = ib.hi;

    async_wait_lds<4>();                          // this wave's pieces of K tile 0 (the B halves of K tile 1 may be in flight)
    lds_barrier();
    if (grp == 1) lds_barrier();                  // group 1 runs one interval behind

    hu32x2 fal[2][4], fah[2][4], fb0l[4], fb0h[4], fb1l[4], fb1h[4];       // low / high half (k 0..3 / 4..7) of each operand
    auto bn_frag = [&](hu32x2& lo, hu32x2& hi2, int u) {                   // BatchNorm + activation on the 8 values of a B fragment
        if constexpr (BNIN) {
            float v[8];
            const hu32x4 w = {lo[0], lo[1], hi2[0], hi2[1]};
            unpack8(w, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bn_act_load(v[e], bsc[u], bsh[u], bneg, bhi);
            const hu32x4 o = pack8(v);
            lo[0] = o[0]; lo[1] = o[1]; hi2[0] = o[2]; hi2[1] = o[3];
        }
    };
    // 8 MFMAs of one quadrant; the two pieces of half-tile (SKT, SH) go behind the 2nd and the 6th; HOOK(s2) runs behind the first MFMA of
    // every K step (P1 uses it for the BatchNorm of b1's fragments: VALU work in the shadow of this wave's own MFMAs instead of in an L
    // section, where it would be time the partner's 8 MFMAs cannot cover)
#define TSII_TNPH_QUADRANT(AT, BU, FBL, FBH, SKT, SH, HOOK)                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                                                  \
        _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                                \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                             \
                const hu32x4 av = {fal[t][s2][0], fal[t][s2][1], fah[t][s2][0], fah[t][s2][1]};                                         \
                const hu32x4 bv = {FBL[s2][0], FBL[s2][1], FBH[s2][0], FBH[s2][1]};                                                     \
                acc[AT + t][BU] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, av), __builtin_bit_cast(hbf16x8, bv), \
                                                                          acc[AT + t][BU], 0, 0, 0);                                    \
                if (t == 0) { HOOK(s2); }                                                                                               \
                if (s2 == 0 && t == 1) { __builtin_amdgcn_sched_barrier(0); stage_piece(SKT, SH, 0); __builtin_amdgcn_sched_barrier(0); } \
                if (s2 == 2 && t == 1) { __builtin_amdgcn_sched_barrier(0); stage_piece(SKT, SH, 1); __builtin_amdgcn_sched_barrier(0); } \
            }                                                                                                                           \
        __builtin_amdgcn_s_setprio(0);
#define TSII_TNPH_NOHOOK(S2)
#define TSII_TNPH_BN1(S2) bn_frag(fb1l[S2], fb1h[S2], 1)
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned char* Sb = S0 + (kt & 1) * BUF_BYTES;
        // ---- P1: a0 (tiles 0, 1), b0 and b1 --------------------------------------------------------------------------------------
        // (lgkmcnt counts to 15: b0 is waited for with one tile of a0 behind it, the other 16 reads fly under b0's BatchNorm)
#define TSII_TNPH_READ4(L, H, BASE)                                                                                       \
        lds_read8_tr<0>(L[0], BASE); lds_read8_tr<4 * 256>(H[0], BASE); lds_read8_tr<16 * 256>(L[1], BASE); lds_read8_tr<20 * 256>(H[1], BASE); \
        lds_read8_tr<32 * 256>(L[2], BASE); lds_read8_tr<36 * 256>(H[2], BASE); lds_read8_tr<48 * 256>(L[3], BASE); lds_read8_tr<52 * 256>(H[3], BASE);
        TSII_TNPH_READ4(fb0l, fb0h, Sb + offB[0])
        TSII_TNPH_READ4(fal[0], fah[0], Sb + offA[0])
        lds_wait4<8>(fb0l[0], fb0l[1], fb0l[2], fb0l[3]); lds_wait4<8>(fb0h[0], fb0h[1], fb0h[2], fb0h[3]);
        TSII_TNPH_READ4(fal[1], fah[1], Sb + offA[1])
        TSII_TNPH_READ4(fb1l, fb1h, Sb + offB[1])
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) bn_frag(fb0l[s2], fb0h[s2], 0);
        lds_wait4<0>(fal[0][0], fal[0][1], fal[0][2], fal[0][3]); lds_wait4<0>(fah[0][0], fah[0][1], fah[0][2], fah[0][3]);
        lds_wait4<0>(fal[1][0], fal[1][1], fal[1][2], fal[1][3]); lds_wait4<0>(fah[1][0], fah[1][1], fah[1][2], fah[1][3]);
        lds_wait4<0>(fb1l[0], fb1l[1], fb1l[2], fb1l[3]); lds_wait4<0>(fb1h[0], fb1h[1], fb1h[2], fb1h[3]);
        lds_barrier();
        TSII_TNPH_QUADRANT(0, 0, fb0l, fb0h, kt + 1, 0, TSII_TNPH_BN1)
        lds_barrier();
        // ---- P2: nothing to read ---------------------------------------------------------------------------------------------------
        lds_barrier();
        TSII_TNPH_QUADRANT(0, 1, fb1l, fb1h, kt + 1, 1, TSII_TNPH_NOHOOK)
        lds_barrier();
        // ---- P3: a1 (tiles 2, 3) over a0 ----------------------------------------------------------------------------------------
        TSII_TNPH_READ4(fal[0], fah[0], Sb + offA[2])
        TSII_TNPH_READ4(fal[1], fah[1], Sb + offA[3])
        lds_wait4<0>(fal[0][0], fal[0][1], fal[0][2], fal[0][3]); lds_wait4<0>(fah[0][0], fah[0][1], fah[0][2], fah[0][3]);
        lds_wait4<0>(fal[1][0], fal[1][1], fal[1][2], fal[1][3]); lds_wait4<0>(fah[1][0], fah[1][1], fah[1][2], fah[1][3]);
        lds_barrier();
        TSII_TNPH_QUADRANT(2, 1, fb1l, fb1h, kt + 2, 2, TSII_TNPH_NOHOOK)
        lds_barrier();
        // ---- P4 ------------------------------------------------------------------------------------------------------------------
        async_wait_lds<2>();                      // everything up to A1 of K tile kt + 1 has landed; B0 of kt + 2 may be in flight
        lds_barrier();
        TSII_TNPH_QUADRANT(2, 0, fb0l, fb0h, kt + 2, 3, TSII_TNPH_NOHOOK)
        lds_barrier();
    }
#undef TSII_TNPH_BN1
#undef TSII_TNPH_READ4
#undef TSII_TNPH_NOHOOK
#undef TSII_TNPH_QUADRANT
    if (grp == 0) lds_barrier();
    async_wait_lds<0>();

    float* Cz = Cws + (int64_t)zsplit * Pn * Q;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (p >= Pn) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + wn * 64 + u * 32 + li;
                if (q < Q) Cz[(int64_t)p * Q + q] = acc[t][u][r];
            }
        }
}

// ---- weights: fp32 reference layout -> the bf16 B operand of the NT kernel, once per call ----------------------------------
//   mode 0 (1x1 forward)   out[n][k]              = w[n][k]
//   mode 1 (1x1 dX)        out[k][n]              = w[n][k]
//   mode 2 (dense forward) out[co][t*cin + ci]    = w[co][ci][t]
//   mode 3 (dense dX)      out[ci][t*cout + co]   = w[co][ci][t]
__global__ void hprep_w_kernel(const float* __restrict__ w, int cout, int cin, int T, int mode, bf16_t* __restrict__ out) {
    const int64_t total = (int64_t)cout * cin * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t src;
        if (mode == 0) src = i;
        else if (mode == 1) { const int k = (int)(i / cout), n = (int)(i % cout); src = (int64_t)n * cin + k; }
        else if (mode == 2) { const int co = (int)(i / ((int64_t)T * cin)); const int r = (int)(i % ((int64_t)T * cin)); const int t = r / cin, ci = r % cin; src = ((int64_t)co * cin + ci) * T + t; }
        else { const int ci = (int)(i / ((int64_t)T * cout)); const int r = (int)(i % ((int64_t)T * cout)); const int t = r / cout, co = r % cout; src = ((int64_t)co * cin + ci) * T + t; }
        out[i] = bf16_bits(w[src]);
    }
}

// column sums of a bf16 [M, N] matrix (bias gradients): partial rows, then the shared row reduction
__global__ __launch_bounds__(256) void hcolsum_kernel(const bf16_t* __restrict__ a, int64_t M, int N, int R, float* __restrict__ part) {
    const int G = N / 8;
    const int64_t tasks = (int64_t)R * G;
    for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < tasks; task += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(task % G) * 8;
        const int r = (int)(task / G);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int64_t m = r; m < M; m += R) {
            float v[8];
            unpack8(ld8(a + m * N + c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(int64_t)r * N + c + e] = s[e];
    }
}

static const HGather kNoHGather = {0, 0, 8, 1, 1, 1, 1, 1, 0, 0, 1, 1};

static inline bf16_t* ws_align16(void* ws) { return reinterpret_cast<bf16_t*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15); }

template <int WM, int WN, int TM, int TN>
static int launch_hnt_cfg(int amode, const bf16_t* A, int64_t lda, const bf16_t* B, bf16_t* C, int64_t ldc, int64_t M, int N, int K,
                          const HEpi& ep, const InBN& ib, const HGather& cg, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "bf16 gemm_nt: grid too large");
    const dim3 grid((unsigned)nblocks);
#define TSII_HNT(AM, BI, BB) hipLaunchKernelGGL((hgemm_nt_kernel<WM, WN, TM, TN, AM, BI, BB>), grid, dim3(256), 0, st, A, lda, B, C, ldc, M, N, K, ep, ntn, ib, cg)
    if (amode == 1) TSII_HNT(1, false, false);
    else if (amode == 2) TSII_HNT(2, false, false);
    else if (ep.bn_y != nullptr) TSII_HNT(0, false, true);
    else if (ib.sc != nullptr) TSII_HNT(0, true, false);
    else TSII_HNT(0, false, false);
#undef TSII_HNT
    return check_launch("bf16 gemm_nt");
}

// A: [M, K] bf16 (amode 0, lda) or gathered by cg (1 / 2); B: bf16 [N, K]; C: bf16 [M, N] (ldc)
static int launch_hnt(int amode, const bf16_t* A, int64_t lda, const bf16_t* B, bf16_t* C, int64_t ldc, int64_t M, int N, int K,
                      const HEpi& ep, const InBN& ib, const HGather& cg, hipStream_t st) {
    TSII_REQUIRE(K % 8 == 0 && K >= 8 && N % 8 == 0 && ldc % 8 == 0 && (amode != 0 || lda % 8 == 0), "bf16 gemm_nt: K, N and the row strides must be multiples of 8");
    TSII_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C), "bf16 gemm_nt: operands must be 16-byte aligned");
    TSII_REQUIRE(!(ep.bn_y != nullptr && (ib.sc != nullptr || amode != 0 || ldc != N)), "bf16 gemm_nt: the BatchNorm-backward epilogue is a plain-operand form");
    // Which kernel (measured on MI355X, tools/bf16_bench.py, profiles/r05n_nt_forms.log, r05q_nt_ph.log):
    //   plain operands, N and K >= 256, and the forward taps of the dense convolutions with cout % 256 == 0, cin % 64 == 0: the
    //       256 x 256 direct-to-LDS kernel (1.1 .. 1.2 PF/s on 65536 x 4096 x 4096 against 0.68 for the 128-row kernel;
    //       131072 x 512 x 512: 88 vs 97 .. 106 us);
    //   the fused forms (BatchNorm-on-load, K6c epilogue) and the other gathered operands of the dense convolutions, N % 256 == 0: the
    //       register-staged kernel with 128 x 256 tiles (wave tile 64 x 128: 0.75 KB of fragment reads per MFMA instead of 1), two
    //       blocks per CU -- 512 -> 256 3x3 forward 465 -> 383 us, K6c dX 131072 x 512 x 512 186 -> 158 us; the fragment-time
    //       BatchNorm of the direct-to-LDS kernels is VALU-bound (218 us) and not instantiated;
    //   everything else: 128 x {128, 64, 32} tiles.
#ifndef HNT_DL
#define HNT_DL 1         // 0: no 256-row kernels (A/B, tools/variants)
#endif
#ifndef HNT_WIDE
#define HNT_WIDE 1       // 0: no 128 x 256 tiles
#endif
    const bool fused = ib.sc != nullptr || ep.bn_y != nullptr;
    if (HNT_DL && amode == 0 && !fused && N >= 256 && M >= 256 && K >= 256) {
        const unsigned ntn = (unsigned)cdiv(N, 256);
        const int64_t nblocks = cdiv64(M, 256) * ntn;
        TSII_REQUIRE(nblocks < (1ll << 31), "bf16 gemm_nt: grid too large");
        hipLaunchKernelGGL((hgemm_nt_ph_kernel<0, false>), dim3((unsigned)nblocks), dim3(512), 0, st, A, lda, B, C, ldc, M, N, K, ep, ntn, cg);
        return check_launch("bf16 gemm_nt (256-row tiles)");
    }
#ifndef HNT_PH_CONV
#define HNT_PH_CONV 1    // 0: the gathered operands of the dense convolutions stay on the 128 x 256 register-staged tiles (A/B)
#endif
    // forward taps only: 3x3 512 -> 256 at 128^2 361-429 us against 402-476 on the 128 x 256 tiles; the dX taps measured 6-10 % SLOWER
    // here (535-581 vs 505-527 us, profiles/r05ag_ph_conv.log) and stay there (HNT_PH_CONV=2 sends them here too)
    if (HNT_DL && HNT_PH_CONV && (amode == 1 || (amode == 2 && HNT_PH_CONV == 2)) && !fused && N % 256 == 0 && M >= 256 && cg.c % 64 == 0 &&
        cg.rh < 32768 && cg.rw < 65536) {
        const unsigned ntn = (unsigned)cdiv(N, 256);
        const dim3 grid((unsigned)(cdiv64(M, 256) * ntn));
        if (amode == 1) hipLaunchKernelGGL((hgemm_nt_ph_kernel<1, false>), grid, dim3(512), 0, st, A, lda, B, C, ldc, M, N, K, ep, ntn, cg);
        else hipLaunchKernelGGL((hgemm_nt_ph_kernel<2, false>), grid, dim3(512), 0, st, A, lda, B, C, ldc, M, N, K, ep, ntn, cg);
        return check_launch("bf16 gemm_nt (256-row tiles, gathered)");
    }
#ifdef HNT_PH_BNB         // A/B: the K6c epilogue behind the quadrant-phase kernel instead of the 128 x 256 register-staged one
    if (amode == 0 && ep.bn_y != nullptr && ib.sc == nullptr && N >= 256 && M >= 256 && K >= 256) {
        const unsigned ntn = (unsigned)cdiv(N, 256);
        hipLaunchKernelGGL((hgemm_nt_ph_kernel<0, true>), dim3((unsigned)(cdiv64(M, 256) * ntn)), dim3(512), 0, st, A, lda, B, C, ldc, M, N, K, ep, ntn, cg);
        return check_launch("bf16 gemm_nt (256-row tiles, K6c)");
    }
#endif
    if (HNT_WIDE && N % 256 == 0 && (amode != 0 || ep.bn_y != nullptr || (ib.sc != nullptr && K >= 256 && K <= HNT_LBN_MAXK)))
        return launch_hnt_cfg<2, 2, 2, 4>(amode, A, lda, B, C, ldc, M, N, K, ep, ib, cg, st);
    if (N % 128 == 0 || N > 192) return launch_hnt_cfg<2, 2, 2, 2>(amode, A, lda, B, C, ldc, M, N, K, ep, ib, cg, st);
    if (N > 32) return launch_hnt_cfg<2, 2, 2, 1>(amode, A, lda, B, C, ldc, M, N, K, ep, ib, cg, st);
    return launch_hnt_cfg<4, 1, 1, 1>(amode, A, lda, B, C, ldc, M, N, K, ep, ib, cg, st);
}

// split of the M rows of a weight-gradient product into chunks (multiples of the 64-row stage): enough blocks to fill the chip.
// ph: the 256 x 256 kernel (one 512-thread block per CU: never more blocks than CUs, a second round would run at a fraction of the chip)
#ifndef HTN_PH
#define HTN_PH 1         // 0: every weight gradient on the register-transposing kernel (A/B, tools/variants)
#endif
static bool htn_use_ph(int Pn, int Q, bool bconv, int cin) {
    return HTN_PH && Pn >= 256 && Pn % 128 == 0 && Q >= 256 && (!bconv || cin % 128 == 0);
}
static void htn_plan(int64_t M, int Pn, int Q, bool ph, int64_t* chunk, int* splits) {
    const int64_t stages = cdiv64(M, HTN_STAGE);
    int64_t want;
    if (ph) {
        const int64_t tiles = (int64_t)cdiv(Pn, 256) * cdiv(Q, 256);
        want = 256 / tiles;
    } else {
        const int64_t tiles = (int64_t)cdiv(Pn, 128) * cdiv(Q, 128);
        want = cdiv64(768, tiles);
        if (want > 256) want = 256;
    }
    if (want > stages) want = stages;
    if (want < 1) want = 1;
    const int64_t per = cdiv64(stages, want);
    *chunk = per * HTN_STAGE;
    *splits = (int)cdiv64(M, *chunk);
}
static size_t htn_ws_floats(int64_t M, int Pn, int Q, bool bconv, int cin) {
    int64_t chunk; int splits;
    htn_plan(M, Pn, Q, htn_use_ph(Pn, Q, bconv, cin), &chunk, &splits);
    return (size_t)splits * Pn * Q;
}
// slabs of partial [Pn, Q] products in ws (htn_ws_floats floats); returns the number of slabs in *nslabs
static int launch_htn(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, bool bconv, const HGather& cg, float* ws, int64_t M, int Pn, int Q,
                      const InBN& ib, int* nslabs, hipStream_t st) {
    TSII_REQUIRE(Pn % 8 == 0 && Q % 8 == 0 && lda % 8 == 0 && (bconv || ldb % 8 == 0) && aligned16(A) && aligned16(B), "bf16 gemm_tn: channel counts must be multiples of 8, operands 16-byte aligned");
    TSII_REQUIRE(M > 0 && M < (1ll << 40), "bf16 gemm_tn: bad row count");
    const bool ph = htn_use_ph(Pn, Q, bconv, cg.c);
    int64_t chunk; int splits;
    htn_plan(M, Pn, Q, ph, &chunk, &splits);
    *nslabs = splits;
    if (ph) {
        const unsigned qt = (unsigned)cdiv(Q, 256), pt = (unsigned)cdiv(Pn, 256);
        const dim3 grid(qt * pt * (unsigned)splits);
        if (bconv) hipLaunchKernelGGL((hgemm_tn_ph_kernel<false, true>), grid, dim3(512), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
        else if (ib.sc != nullptr) hipLaunchKernelGGL((hgemm_tn_ph_kernel<true, false>), grid, dim3(512), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
        else hipLaunchKernelGGL((hgemm_tn_ph_kernel<false, false>), grid, dim3(512), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
        return check_launch("bf16 gemm_tn (256 x 256 tiles)");
    }
    const unsigned qt = (unsigned)cdiv(Q, 128), pt = (unsigned)cdiv(Pn, 128);
    const int64_t nblocks = (int64_t)qt * pt * splits;
    TSII_REQUIRE(nblocks < (1ll << 31), "bf16 gemm_tn: grid too large");
    const dim3 grid((unsigned)nblocks);
    if (bconv) hipLaunchKernelGGL((hgemm_tn_kernel<false, true>), grid, dim3(256), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
    else if (ib.sc != nullptr) hipLaunchKernelGGL((hgemm_tn_kernel<true, false>), grid, dim3(256), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
    else hipLaunchKernelGGL((hgemm_tn_kernel<false, false>), grid, dim3(256), 0, st, A, lda, B, ldb, ws, M, Pn, Q, chunk, ib, qt, pt, cg);
    return check_launch("bf16 gemm_tn");
}

static int prep_weights(const float* w, int cout, int cin, int T, int mode, bf16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(hprep_w_kernel, dim3(stream_grid((int64_t)cout * cin * T, 256)), dim3(256), 0, st, w, cout, cin, T, mode, out);
    return check_launch("bf16 prep_w");
}

static inline int hcolsum_rows(int64_t M, int N) { return partial_rows(M, N / 8); }
int launch_bf16_colsum(const bf16_t* a, int64_t M, int N, float* out, float* ws, hipStream_t st) {
    const int R = hcolsum_rows(M, N);
    hipLaunchKernelGGL(hcolsum_kernel, dim3(stream_grid((int64_t)R * (N / 8), 256)), dim3(256), 0, st, a, M, N, R, ws);
    int rc = check_launch("bf16 colsum");
    if (rc) return rc;
    return launch_reduce_rows(ws, R, N, out, st);
}

static int make_inbn_checked(const float* sc, const float* sh, int act, float slope, InBN* ib, const char* who) {
    if (sc == nullptr) { ib->sc = nullptr; ib->sh = nullptr; ib->neg = 1.f; ib->hi = __builtin_huge_valf(); return 0; }
    TSII_REQUIRE(sh != nullptr && aligned16(sc) && aligned16(sh), "%s: in_scale / in_shift go together and must be 16-byte aligned", who);
    TSII_REQUIRE(make_in_bn(sc, sh, act, slope, ib) == 0, "%s: activation %d (slope %g) has no load-time form", who, act, (double)slope);
    return 0;
}

static HGather make_gather(int h, int w, int c, int rh, int rw, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    HGather g = {h, w, c, rh, rw, kw, sh, sw, ph, pw, dh, dw};
    return g;
}

}  // namespace tsii

using namespace tsii;

// ================================================================================================================================
// C ABI (include/tsii_hip.h, "bf16 activation storage")
// ================================================================================================================================
extern "C" int64_t tsii_bf16_stat_rows(int64_t m) { return m > 0 ? cdiv64(m, 128) : 0; }

extern "C" size_t tsii_bf16_pw_ws_bytes(int n, int k) { return (n > 0 && k > 0) ? (size_t)n * k * sizeof(bf16_t) + 16 : 0; }

extern "C" int tsii_bf16_pw_fwd(const uint16_t* x, int64_t m, int k, const float* w, int n, const float* bias,
                                const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                float* stat_part, uint16_t* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && w && y && ws, "bf16_pw_fwd: null pointer");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0 && k % 8 == 0 && n % 8 == 0, "bf16_pw_fwd: channel counts must be positive multiples of 8 (got k=%d n=%d)", k, n);
    TSII_REQUIRE(ws_bytes >= tsii_bf16_pw_ws_bytes(n, k), "bf16_pw_fwd: workspace too small (tsii_bf16_pw_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    InBN ib;
    if (make_inbn_checked(in_scale, in_shift, in_act, in_slope, &ib, "bf16_pw_fwd")) return -1;
    bf16_t* wb = ws_align16(ws);
    int rc = prep_weights(w, n, k, 1, 0, wb, st);
    if (rc) return rc;
    HEpi ep = {bias, stat_part, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f, nullptr};
    return launch_hnt(0, x, k, wb, y, n, m, n, k, ep, ib, kNoHGather, st);
}

extern "C" int tsii_bf16_pw_bwd_dx(const uint16_t* dy, int64_t m, int n, const float* w, int k,
                                   const uint16_t* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                   const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                   uint16_t* dx, float* bwd_part, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && w && dx && ws, "bf16_pw_bwd_dx: null pointer");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0 && k % 8 == 0 && n % 8 == 0, "bf16_pw_bwd_dx: channel counts must be positive multiples of 8");
    TSII_REQUIRE(ws_bytes >= tsii_bf16_pw_ws_bytes(n, k), "bf16_pw_bwd_dx: workspace too small (tsii_bf16_pw_ws_bytes)");
    TSII_REQUIRE((bn_y == nullptr) == (bwd_part == nullptr), "bf16_pw_bwd_dx: bn_y and bwd_part go together");
    hipStream_t st = (hipStream_t)stream;
    bf16_t* wb = ws_align16(ws);
    int rc = prep_weights(w, n, k, 1, 1, wb, st);       // [k][n]
    if (rc) return rc;
    HEpi ep = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f, nullptr};
    InBN none = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    if (bn_y != nullptr) {
        TSII_REQUIRE(bn_mean && bn_var && bn_gamma && bn_beta, "bf16_pw_bwd_dx: BatchNorm parameters missing");
        InBN tmp;
        TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "bf16_pw_bwd_dx: activation %d has no load-time form", bn_act);
        ep.bn_y = bn_y; ep.bn_mean = bn_mean; ep.bn_var = bn_var; ep.bn_gamma = bn_gamma; ep.bn_beta = bn_beta;
        ep.bn_eps = bn_eps; ep.bn_neg = tmp.neg; ep.bn_hi = tmp.hi; ep.bn_part = bwd_part;
    }
    return launch_hnt(0, dy, n, wb, dx, k, m, k, n, ep, none, kNoHGather, st);
}

extern "C" size_t tsii_bf16_pw_bwd_dw_ws_bytes(int64_t m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const size_t slabs = htn_ws_floats(m, n, k, false, 0);
    const size_t bias = (size_t)hcolsum_rows(m, n) * n;
    return (slabs > bias ? slabs : bias) * sizeof(float) + 16;
}

extern "C" int tsii_bf16_pw_bwd_dw(const uint16_t* dy, const uint16_t* x, int64_t m, int n, int k,
                                   const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                   float* dw, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && x && dw && ws, "bf16_pw_bwd_dw: null pointer");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0 && k % 8 == 0 && n % 8 == 0, "bf16_pw_bwd_dw: channel counts must be positive multiples of 8");
    TSII_REQUIRE(ws_bytes >= tsii_bf16_pw_bwd_dw_ws_bytes(m, n, k), "bf16_pw_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    InBN ib;
    if (make_inbn_checked(in_scale, in_shift, in_act, in_slope, &ib, "bf16_pw_bwd_dw")) return -1;
    float* wsf = reinterpret_cast<float*>(ws_align16(ws));
    int slabs = 0;
    int rc = launch_htn(dy, n, x, k, false, kNoHGather, wsf, m, n, k, ib, &slabs, st);
    if (rc) return rc;
    rc = launch_reduce_rows(wsf, slabs, (int64_t)n * k, dw, st);
    if (rc) return rc;
    if (dbias != nullptr) return launch_bf16_colsum(dy, m, n, dbias, wsf, st);
    return 0;
}

// ---- dense k x k convolutions as implicit GEMM ------------------------------------------------------------------------------
extern "C" size_t tsii_bf16_dense_ws_bytes(int cin, int cout, int kh, int kw) {
    return (cin > 0 && cout > 0 && kh > 0 && kw > 0) ? (size_t)cin * cout * kh * kw * sizeof(bf16_t) + 16 : 0;
}

static int dense_geom_ok(const char* who, int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo) {
    TSII_REQUIRE(n > 0 && h > 0 && wd > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "%s: bad geometry", who);
    TSII_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "%s: channel counts must be multiples of 8 (got %d -> %d); pad on the host side", who, cin, cout);
    TSII_REQUIRE(ho == (h + 2 * ph - dh * (kh - 1) - 1) / sh + 1 && wo == (wd + 2 * pw - dw * (kw - 1) - 1) / sw + 1, "%s: output size does not match the geometry", who);
    TSII_REQUIRE((int64_t)n * h * wd < (1ll << 31) && (int64_t)n * ho * wo < (1ll << 31), "%s: more than 2^31 pixels", who);
    return 0;
}

extern "C" int tsii_bf16_dense_fwd(const uint16_t* x, const float* w, const float* bias, int n, int h, int wd, int cin, int cout,
                                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                   float* stat_part, uint16_t* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && w && y && ws, "bf16_dense_fwd: null pointer");
    if (dense_geom_ok("bf16_dense_fwd", n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_bf16_dense_ws_bytes(cin, cout, kh, kw), "bf16_dense_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    bf16_t* wb = ws_align16(ws);
    int rc = prep_weights(w, cout, cin, kh * kw, 2, wb, st);
    if (rc) return rc;
    const HGather cg = make_gather(h, wd, cin, ho, wo, kw, sh, sw, ph, pw, dh, dw);
    HEpi ep = {bias, stat_part, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f, nullptr};
    InBN none = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    return launch_hnt(1, x, 0, wb, y, cout, (int64_t)n * ho * wo, cout, kh * kw * cin, ep, none, cg, st);
}

extern "C" int tsii_bf16_dense_bwd_dx(const uint16_t* dy, const float* w, int n, int h, int wd, int cin, int cout,
                                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                      uint16_t* dx, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && w && dx && ws, "bf16_dense_bwd_dx: null pointer");
    if (dense_geom_ok("bf16_dense_bwd_dx", n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_bf16_dense_ws_bytes(cin, cout, kh, kw), "bf16_dense_bwd_dx: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    bf16_t* wb = ws_align16(ws);
    int rc = prep_weights(w, cout, cin, kh * kw, 3, wb, st);      // [ci][t * cout + co]
    if (rc) return rc;
    // rows = input pixels, source = dy [n, ho, wo, cout]
    const HGather cg = make_gather(ho, wo, cout, h, wd, kw, sh, sw, ph, pw, dh, dw);
    HEpi ep = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1.f, 0.f, nullptr};
    InBN none = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    return launch_hnt(2, dy, 0, wb, dx, cin, (int64_t)n * h * wd, cin, kh * kw * cout, ep, none, cg, st);
}

extern "C" size_t tsii_bf16_dense_bwd_dw_ws_bytes(int n, int ho, int wo, int cin, int cout, int kh, int kw) {
    if (n <= 0 || ho <= 0 || wo <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0) return 0;
    const int64_t m = (int64_t)n * ho * wo;
    const size_t slabs = htn_ws_floats(m, cout, kh * kw * cin, true, cin);
    const size_t bias = (size_t)hcolsum_rows(m, cout) * cout;
    return (slabs > bias ? slabs : bias) * sizeof(float) + 16;
}

extern "C" int tsii_bf16_dense_bwd_dw(const uint16_t* dy, const uint16_t* x, int n, int h, int wd, int cin, int cout,
                                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                      float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && x && dwgt && ws, "bf16_dense_bwd_dw: null pointer");
    if (dense_geom_ok("bf16_dense_bwd_dw", n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo)) return -1;
    TSII_REQUIRE(ws_bytes >= tsii_bf16_dense_bwd_dw_ws_bytes(n, ho, wo, cin, cout, kh, kw), "bf16_dense_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t m = (int64_t)n * ho * wo;
    const HGather cg = make_gather(h, wd, cin, ho, wo, kw, sh, sw, ph, pw, dh, dw);
    InBN none = {nullptr, nullptr, 1.f, __builtin_huge_valf()};
    float* wsf = reinterpret_cast<float*>(ws_align16(ws));
    int slabs = 0;
    int rc = launch_htn(dy, cout, x, 0, true, cg, wsf, m, cout, kh * kw * cin, none, &slabs, st);
    if (rc) return rc;
    rc = launch_reduce_rows_conv(wsf, slabs, cout, cin, kh * kw, dwgt, st);      // [co][t][ci] partials -> [co][ci][t]
    if (rc) return rc;
    if (dbias != nullptr) return launch_bf16_colsum(dy, m, cout, dbias, wsf, st);
    return 0;
}
