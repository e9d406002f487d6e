// K3: point-wise (1x1) partial convolution as an fp32 GEMM on the gfx950 matrix cores.
//
// NHWC activations make a 1x1 conv a row-major GEMM  Y[M,N] = X[M,K] * W[N,K]^T  with
// M = N_img*H*W (up to 2M rows at 512^2 bs 32) and K,N in 32..1024.  fp32 parity (1e-3) rules
// out bf16 inputs, so the kernels use v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF/s
// peak = the fp32 vector peak but with one VGPR per operand and the VALU left free).
//
//   gemm_nt : C[M,N] = sum_k A[m,k]*rs(m,k) * B[n,k]      forward, and dX with B = W^T
//   gemm_tn : C[P,Q] = sum_m A[m,p]*sa[m]  * B[m,q]*rs(m,q)   dW, split over m (partials + reduce)
//
// Block = 256 threads = 4 waves; 128x128 (or 128x64 / 128x32) output tile, BK = 32; A/B tiles
// are staged global -> registers -> LDS with the next tile's global loads issued before the
// MFMA phase of the current one.  LDS rows are padded to 36 floats so the ds_read_b128 fragment
// reads (16-lane groups, 144-byte row stride) are bank-conflict free.  The mask multiply
// (x*m, partial_convolution.py:51/123), the division by the valid count and the hole zeroing
// (:66-72) ride in the tile loader / epilogue, so no masked copy of x is ever materialised.
#include "tsii_common.h"
#include "gemm_tiles.h"

namespace tsii {

template <int WM, int WN, int TM, int TN, bool VEC, int AMODE, bool BNIN = false, bool BNB = false>
__global__ __launch_bounds__(256, BNB ? 2 : 3) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda, RowScale as,
                                                      const float* __restrict__ B, int64_t ldb,
                                                      float* __restrict__ C, int64_t ldc,
                                                      int64_t M, int N, int K, Epilogue ep, unsigned ntn,
                                                      ConvGather cg, InBN ib) {
    static_assert(!BNIN || (VEC && AMODE == 0), "input BatchNorm rides the plain vector loader");
    static_assert(!BNB || (VEC && AMODE == 0 && !BNIN), "K6c rides the plain dX form");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * GEMM_LDS];
    float* As = smem;
    float* Bs = smem + BM * GEMM_LDS;

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

    float4 ra[BM / 32], rb[BN / 32];
    float sa0[BM / 32], sa1[BM / 32], sb0[BN / 32], sb1[BN / 32];
    int rn[BM / 32], ry[BM / 32], rx[BM / 32];
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) { sb0[i] = 1.f; sb1[i] = 1.f; }
    if constexpr (AMODE == 0) {
        nt_row_scales<BM>(as, m0, M, sa0, sa1);
        nt_load<BM, VEC>(A, lda, m0, M, 0, K, ra);
    } else {
        conv_rows<BM>(cg, m0, M, rn, ry, rx);
        conv_load<BM, AMODE>(A, cg, rn, ry, rx, 0, K, ra, sa0, sa1);
    }
    nt_load<BN, VEC>(B, ldb, n0, N, 0, K, rb);
    float4 psc = make_float4(0.f, 0.f, 0.f, 0.f), psh = psc;   // BNIN: this thread's 4 channels of the tile in flight
    const int pk = (tid & 7) * 4;
    if constexpr (BNIN) {
        if (pk < K) { psc = *reinterpret_cast<const float4*>(ib.sc + pk); psh = *reinterpret_cast<const float4*>(ib.sh + pk); }
        nt_store_bn<BM>(As, ra, 0, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi);
    } else if constexpr (AMODE == 0) nt_store<BM, true>(As, ra, 0, as.split, sa0, sa1);
    else conv_store<BM>(As, ra, cg, 0, sa0, sa1);
    nt_store<BN, false>(Bs, rb, 0, 0, sb0, sb1);
    __syncthreads();

    const int nk = (K + GEMM_BK - 1) / GEMM_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {  // next tile's global loads fly during the MFMA phase
            if constexpr (AMODE == 0) nt_load<BM, VEC>(A, lda, m0, M, (kt + 1) * GEMM_BK, K, ra);
            else conv_load<BM, AMODE>(A, cg, rn, ry, rx, (kt + 1) * GEMM_BK, K, ra, sa0, sa1);
            nt_load<BN, VEC>(B, ldb, n0, N, (kt + 1) * GEMM_BK, K, rb);
            if constexpr (BNIN) {
                const int k = (kt + 1) * GEMM_BK + pk;
                psc = make_float4(0.f, 0.f, 0.f, 0.f); psh = psc;
                if (k < K) { psc = *reinterpret_cast<const float4*>(ib.sc + k); psh = *reinterpret_cast<const float4*>(ib.sh + k); }
            }
        }
        // (tried: fragments as double-buffered 8-byte reads issued 8 MFMAs ahead -- 2-way bank conflicts and twice the LDS
        // instructions cost more than the hidden latency: 16.9 -> 17.6 ms over the forward GEMMs)
        __builtin_amdgcn_s_setprio(2);   // MFMA phase first: the other waves' load / store phases fill the gaps
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a[TM][4], b[TN][4];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(As + ((wm * TM + t) * 32 + li) * GEMM_LDS + 8 * s + 4 * hi);
                a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
            }
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + ((wn * TN + u) * 32 + li) * GEMM_LDS + 8 * s + 4 * hi);
                b[u][0] = v.x; b[u][1] = v.y; b[u][2] = v.z; b[u][3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < TN; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[u][j], acc[t][u], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (more) {
            if constexpr (BNIN) nt_store_bn<BM>(As, ra, (kt + 1) * GEMM_BK, as.split, sa0, sa1, psc, psh, ib.neg, ib.hi);
            else if constexpr (AMODE == 0) nt_store<BM, true>(As, ra, (kt + 1) * GEMM_BK, as.split, sa0, sa1);
            else conv_store<BM>(As, ra, cg, (kt + 1) * GEMM_BK, sa0, sa1);
            nt_store<BN, false>(Bs, rb, 0, 0, sb0, sb1);
            __syncthreads();
        }
    }

    nt_epilogue<WM, WN, TM, TN, AMODE, BNB, (BM + BN) * GEMM_LDS>(smem, acc, C, ldc, M, N, ep, cg, m0, n0, bid, ntn);
}

// ---- TN (dW): reduction over rows m, operands are [m][channel] --------------------------
// TN tile: 32 rows (m) x COLS channels.  Loads are raw; the per-row factors (rowmul, and the two-plane
// row scale) are fetched into `f0`/`f1` next to them and applied by tn_store after the MFMA phase.
// Thread (r = tid / 8, j = tid % 8) loads float4 columns j, j + 8, ... of ONE row, so it carries a single set of
// per-row factors; LDS rows are COLS + 32 floats apart so the 16-lane ds_write_b128 groups (two rows of 8
// float4) and the two half-waves of a fragment read (rows k, k + 1) land on disjoint banks.
template <int COLS>
struct TnLd { static constexpr int value = COLS + 32; };

template <int COLS, bool VEC>
__device__ __forceinline__ void tn_load(const float* __restrict__ P, int64_t ld, int64_t m0, int64_t mend,
                                        int c0, int ncols, const float* __restrict__ rowmul, const RowScale& rs,
                                        float4 (&regs)[COLS / 32], float& f0, float& f1) {
    // wave-uniform 64-bit bases (SGPRs) + 32-bit per-thread offsets: 64-bit per-load address VGPRs made this
    // kernel spill, and every spill reload put an s_waitcnt vmcnt(0) in the middle of the load issue
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float* __restrict__ base = P + m0 * ld;
    const int left = (int)((mend - m0 < 32) ? (mend - m0) : 32);
    const bool ok = r < left;
    const float* p = base + (r * (int)ld + c0 + j * 4);
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int c = c0 + (j + 8 * i) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (VEC) {
                if (c < ncols) v = *reinterpret_cast<const float4*>(p + 32 * i);
            } else {
                if (c + 0 < ncols) v.x = p[32 * i + 0];
                if (c + 1 < ncols) v.y = p[32 * i + 1];
                if (c + 2 < ncols) v.z = p[32 * i + 2];
                if (c + 3 < ncols) v.w = p[32 * i + 3];
            }
        }
        regs[i] = v;
    }
    float a0 = 1.f, a1 = 1.f;
    if (ok) {
        if (rowmul != nullptr) { a0 = rowmul[m0 + r]; a1 = a0; }
        if (rs.r0 != nullptr) {
            a0 *= rs.r0[m0 + r];
            a1 *= rs.r1 != nullptr ? rs.r1[m0 + r] : 1.f;
        }
    }
    f0 = a0;
    f1 = a1;
}

template <int COLS>
__device__ __forceinline__ void tn_store(float* __restrict__ S, const float4 (&regs)[COLS / 32], int c0, int split,
                                         bool split_active, float f0, float f1) {
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int c4 = j + 8 * i;
        const int c = c0 + c4 * 4;
        float4 v = regs[i];
        v.x *= (!split_active || c + 0 < split) ? f0 : f1;
        v.y *= (!split_active || c + 1 < split) ? f0 : f1;
        v.z *= (!split_active || c + 2 < split) ? f0 : f1;
        v.w *= (!split_active || c + 3 < split) ? f0 : f1;
        *reinterpret_cast<float4*>(S + r * TnLd<COLS>::value + c4 * 4) = v;
    }
}

// tn_store with the producer's BatchNorm + activation applied first; tab = [sc[COLS] | sh[COLS]] of this block's columns
template <int COLS>
__device__ __forceinline__ void tn_store_bn(float* __restrict__ S, const float4 (&regs)[COLS / 32], int c0, int split,
                                            bool split_active, float f0, float f1, const float* __restrict__ tab, float neg,
                                            float hi) {
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int c4 = j + 8 * i;
        const int c = c0 + c4 * 4;
        const float4 sc = *reinterpret_cast<const float4*>(tab + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4*>(tab + COLS + c4 * 4);
        float4 v = regs[i];
        v.x = bn_act_load(v.x, sc.x, sh.x, neg, hi) * ((!split_active || c + 0 < split) ? f0 : f1);
        v.y = bn_act_load(v.y, sc.y, sh.y, neg, hi) * ((!split_active || c + 1 < split) ? f0 : f1);
        v.z = bn_act_load(v.z, sc.z, sh.z, neg, hi) * ((!split_active || c + 2 < split) ? f0 : f1);
        v.w = bn_act_load(v.w, sc.w, sh.w, neg, hi) * ((!split_active || c + 3 < split) ? f0 : f1);
        *reinterpret_cast<float4*>(S + r * TnLd<COLS>::value + c4 * 4) = v;
    }
}

// gathered B operand of the TN kernel (dW of a dense conv): rows m = output pixels, columns k = (tap, ci) of x
template <int COLS, bool ELEM>
__device__ __forceinline__ void conv_tn_load(const float* __restrict__ src, const ConvGather& cg, int64_t m0, int64_t mend,
                                             int q0, int Q, float4 (&regs)[COLS / 32], float (&f0)[COLS / 32],
                                             float (&f1)[COLS / 32]) {
    const int tid = threadIdx.x;
    const int k = q0 + (tid % (COLS / 4)) * 4;        // 256 % (COLS/4) == 0: one column group per thread
    if constexpr (ELEM) {
#pragma unroll
        for (int i = 0; i < COLS / 32; ++i) {
            const int64_t row = m0 + (tid + 256 * i) / (COLS / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < mend) {
                const int rx = (int)(row % cg.rw), ry = (int)((row / cg.rw) % cg.rh);
                const int n = (int)(row / ((int64_t)cg.rw * cg.rh));
                if (k + 0 < Q) v.x = conv_gather_elem(src, cg, n, ry, rx, k + 0);
                if (k + 1 < Q) v.y = conv_gather_elem(src, cg, n, ry, rx, k + 1);
                if (k + 2 < Q) v.z = conv_gather_elem(src, cg, n, ry, rx, k + 2);
                if (k + 3 < Q) v.w = conv_gather_elem(src, cg, n, ry, rx, k + 3);
            }
            regs[i] = v; f0[i] = 1.f; f1[i] = 1.f;
        }
        return;
    }
    const int t = k / cg.c, ci = k - t * cg.c;
    const int ky = t / cg.kw, kx = t - ky * cg.kw;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int64_t row = m0 + (tid + 256 * i) / (COLS / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float a0 = 1.f, a1 = 1.f;
        if (row < mend && k < Q) {
            const int rx = (int)(row % cg.rw), ry = (int)((row / cg.rw) % cg.rh);
            const int64_t n = row / ((int64_t)cg.rw * cg.rh);
            int sy, sx;
            if (conv_src<1>(cg, ry, rx, ky, kx, sy, sx)) {
                const int64_t spix = (n * cg.h + sy) * cg.w + sx;
                v = *reinterpret_cast<const float4*>(src + spix * cg.c + ci);
                if (cg.p0 != nullptr) { a0 = cg.p0[spix]; a1 = cg.p1 != nullptr ? cg.p1[spix] : 1.f; }
            }
        }
        regs[i] = v; f0[i] = a0; f1[i] = a1;
    }
}
template <int COLS>
__device__ __forceinline__ void conv_tn_store(float* __restrict__ S, const float4 (&regs)[COLS / 32], const ConvGather& cg, int q0,
                                              const float (&f0)[COLS / 32], const float (&f1)[COLS / 32]) {
    const int tid = threadIdx.x;
    const int c4 = tid % (COLS / 4);
    const int ci = (q0 + c4 * 4) % cg.c;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int r = (tid + 256 * i) / (COLS / 4);
        float4 v = regs[i];
        v.x *= (ci + 0 < cg.split) ? f0[i] : f1[i];
        v.y *= (ci + 1 < cg.split) ? f0[i] : f1[i];
        v.z *= (ci + 2 < cg.split) ? f0[i] : f1[i];
        v.w *= (ci + 3 < cg.split) ? f0[i] : f1[i];
        *reinterpret_cast<float4*>(S + r * TnLd<COLS>::value + c4 * 4) = v;
    }
}

template <int WM, int WN, int TM, int TN, bool VEC, int BCONV, bool BNIN = false>
__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ sa,
                                                      const float* __restrict__ B, int64_t ldb, RowScale sb,
                                                      float* __restrict__ Cws, int64_t M, int P, int Q, int64_t chunk,
                                                      ConvGather cg, InBN ib) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(!BNIN || BCONV == 0, "input BatchNorm rides the plain loader");
    __shared__ __attribute__((aligned(16))) float bnt[BNIN ? 2 * BN : 4];
    constexpr int LDA = TnLd<BM>::value, LDB = TnLd<BN>::value;
    __shared__ __attribute__((aligned(16))) float smem[GEMM_BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + GEMM_BK * LDA;

    const int q0 = blockIdx.x * BN, p0 = blockIdx.y * BM;
    const int64_t mbeg = (int64_t)blockIdx.z * chunk;
    const int64_t mend = (mbeg + chunk < M) ? mbeg + chunk : M;

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

    float4 ra[BM / 32], rb[BN / 32];
    float fa0, fa1, fb0, fb1;
    float gb0[BN / 32], gb1[BN / 32];   // gathered B operand: one factor pair per load
    const RowScale none = {nullptr, nullptr, 0};
    const bool sb_active = sb.r0 != nullptr;
    if constexpr (BNIN) {   // (scale, shift) of this block's B columns; columns >= Q stay 0
        if (tid < BN) {
            const int q = q0 + tid;
            bnt[tid] = q < Q ? ib.sc[q] : 0.f;
            bnt[BN + tid] = q < Q ? ib.sh[q] : 0.f;
        }
        __syncthreads();
    }
    tn_load<BM, VEC>(A, lda, mbeg, mend, p0, P, sa, none, ra, fa0, fa1);
    if constexpr (BCONV != 0) conv_tn_load<BN, BCONV == 2>(B, cg, mbeg, mend, q0, Q, rb, gb0, gb1);
    else tn_load<BN, VEC>(B, ldb, mbeg, mend, q0, Q, nullptr, sb, rb, fb0, fb1);
    tn_store<BM>(As, ra, p0, 0, false, fa0, fa1);
    if constexpr (BCONV != 0) conv_tn_store<BN>(Bs, rb, cg, q0, gb0, gb1);
    else if constexpr (BNIN) tn_store_bn<BN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1, bnt, ib.neg, ib.hi);
    else tn_store<BN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1);
    __syncthreads();

    for (int64_t mt = mbeg; mt < mend; mt += GEMM_BK) {
        const bool more = (mt + GEMM_BK < mend);
        if (more) {
            tn_load<BM, VEC>(A, lda, mt + GEMM_BK, mend, p0, P, sa, none, ra, fa0, fa1);
            if constexpr (BCONV != 0) conv_tn_load<BN, BCONV == 2>(B, cg, mt + GEMM_BK, mend, q0, Q, rb, gb0, gb1);
            else tn_load<BN, VEC>(B, ldb, mt + GEMM_BK, mend, q0, Q, nullptr, sb, rb, fb0, fb1);
        }
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[(2 * kk + hi) * LDA + (wm * TM + t) * 32 + li];
#pragma unroll
            for (int u = 0; u < TN; ++u) b[u] = Bs[(2 * kk + hi) * LDB + (wn * TN + u) * 32 + li];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        if (more) {
            tn_store<BM>(As, ra, p0, 0, false, fa0, fa1);
            if constexpr (BCONV != 0) conv_tn_store<BN>(Bs, rb, cg, q0, gb0, gb1);
            else if constexpr (BNIN) tn_store_bn<BN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1, bnt, ib.neg, ib.hi);
            else tn_store<BN>(Bs, rb, q0, sb.split, sb_active, fb0, fb1);
            __syncthreads();
        }
    }

    float* Cz = Cws + (int64_t)blockIdx.z * P * Q;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + (wm * TM + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (p >= P) continue;
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int q = q0 + (wn * TN + u) * 32 + li;
                if (q < Q) Cz[(int64_t)p * Q + q] = acc[t][u][r];
            }
        }
}

// ---- host-side dispatch ----------------------------------------------------------------
static const ConvGather kNoConv = {0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, nullptr, nullptr, 0, nullptr, 0, 0, 0, 0, 0, 0};
static const InBN kNoBN = {nullptr, nullptr, 1.f, 0.f};

template <int WM, int WN, int TM, int TN, int AMODE>
static int launch_nt_conv_cfg(const float* A, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                              Epilogue ep, ConvGather cg, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "conv gemm: grid too large");
    const RowScale none = {nullptr, nullptr, 0};
    hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, true, AMODE>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                       A, (int64_t)0, none, B, ldb, C, ldc, M, N, K, ep, ntn, cg, kNoBN);
    return check_launch("conv_gemm_nt");
}
template <int AMODE>
static int launch_nt_conv(const float* A, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                          Epilogue ep, ConvGather cg, hipStream_t stream) {
    ep.vec_store = (ldc % 4 == 0) && aligned16(C);
    if constexpr (AMODE == 1 || AMODE == 2) {
        if (nt_split_conv_ok(AMODE, A, B, ldb, K, cg)) return launch_nt_split_conv(AMODE, A, B, ldb, C, ldc, M, N, K, ep, cg, stream);
    }
    if (N % 128 == 0 || N > 192) return launch_nt_conv_cfg<2, 2, 2, 2, AMODE>(A, B, ldb, C, ldc, M, N, K, ep, cg, stream);
    if (N > 32) return launch_nt_conv_cfg<2, 2, 2, 1, AMODE>(A, B, ldb, C, ldc, M, N, K, ep, cg, stream);
    return launch_nt_conv_cfg<4, 1, 1, 1, AMODE>(A, B, ldb, C, ldc, M, N, K, ep, cg, stream);
}

template <int WM, int WN, int TM, int TN>
static int launch_nt_cfg(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, float* C, int64_t ldc,
                         int64_t M, int N, int K, Epilogue ep, bool vec, InBN ib, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t nblocks = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(nblocks < (1ll << 31), "gemm_nt: grid too large");
    if (ep.bn_y != nullptr) {
        TSII_REQUIRE(vec && ib.sc == nullptr && N % 4 == 0 && ldc == N && aligned16(ep.bn_y) && ep.vec_store,
                     "gemm_nt: the BatchNorm-backward epilogue needs N %% 4 == 0 and 16-byte aligned operands");
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, true, 0, false, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, kNoConv, kNoBN);
    } else if (ib.sc != nullptr) {
        TSII_REQUIRE(vec && aligned16(ib.sc) && aligned16(ib.sh), "gemm_nt: input BatchNorm needs K %% 4 == 0 and 16-byte aligned operands");
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, true, 0, true>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, kNoConv, ib);
    } else if (vec)
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, true, 0>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, kNoConv, kNoBN);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, false, 0>), dim3((unsigned)nblocks), dim3(256), 0, stream,
                           A, lda, as, B, ldb, C, ldc, M, N, K, ep, ntn, kNoConv, kNoBN);
    return check_launch("gemm_nt");
}

static int launch_nt(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int N, int K, Epilogue ep, hipStream_t stream, InBN ib = kNoBN, void* wsplit = nullptr) {
    if (wsplit != nullptr && ldb == K && nt_split_ok(A, lda, B, ldb, K) && nt_pc_ok(A, lda, as, M, N, K, ep, ib))
        return launch_nt_pc(A, lda, as, B, ldb, false, C, ldc, M, N, K, ep, ib, wsplit, stream);
    if (nt_split_ok(A, lda, B, ldb, K)) return launch_nt_split(A, lda, as, B, ldb, false, C, ldc, M, N, K, ep, ib, wsplit, stream);
    const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A) && aligned16(B);
    ep.vec_store = (ldc % 4 == 0) && aligned16(C);
    if (N % 128 == 0 || N > 192) return launch_nt_cfg<2, 2, 2, 2>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, vec, ib, stream);
    if (N > 32) return launch_nt_cfg<2, 2, 2, 1>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, vec, ib, stream);
    return launch_nt_cfg<4, 1, 1, 1>(A, lda, as, B, ldb, C, ldc, M, N, K, ep, vec, ib, stream);
}

struct TnPlan {
    bool big;
    bool narrow;   // 128 x 64 tiles: Q leaves at most half of its last 128-wide tile (e.g. 384 x 192)
    int bm, bn;
    int splits;
    int64_t chunk;
    bool thin;      // 32 x 128 tiles (split-bf16 kernel only)
};
static TnPlan plan_tn(int64_t M, int P, int Q, bool allow_narrow = false, bool allow_thin = false) {
    TnPlan pl;
    pl.big = (P >= 128 && Q >= 128);
    pl.bm = pl.bn = pl.big ? 128 : 64;
    pl.thin = false;
#ifndef TSII_TN_THIN32
#define TSII_TN_THIN32 1
#endif
    // <= 32 x wide products (the first block's project layer: 2M x 32 x 384): 32 x 128 tiles on the split-bf16 kernel -- the 64 x 64
    // tiles gave two of the four waves rows of padding to multiply and split the 32-channel panel once per 64 columns (round 6)
    if (TSII_TN_THIN32 && allow_thin && allow_narrow && P <= 32 && Q >= 128) {
        pl.thin = true; pl.bm = 32; pl.bn = 128;
        const int tiles_t = cdiv(Q, 128);
        int64_t want_t = 768 / tiles_t;
        if (want_t < 1) want_t = 1;
        int64_t chunk_t = cdiv64(M, want_t);
        if (chunk_t < 256) chunk_t = 256;
        chunk_t = cdiv64(chunk_t, GEMM_BK) * GEMM_BK;
        pl.chunk = chunk_t; pl.splits = (int)cdiv64(M, chunk_t); pl.narrow = false;
        return pl;
    }
    pl.narrow = allow_narrow && pl.big && (Q % 128) >= 1 && (Q % 128) <= 64;
#ifndef TSII_TN_NARROW64
#define TSII_TN_NARROW64 1
#endif
    // tall x 64 products (the 64-channel skip halves of the decoder's 1x1 convs, K7b: 384 x 64, 256 x 64): 128 x 64 tiles read the
    // narrow operand's panel once per 128 rows of the wide one instead of once per 64
    if (TSII_TN_NARROW64 && allow_narrow && P >= 128 && Q > 32 && Q <= 64) { pl.narrow = true; pl.bm = 128; }
    if (pl.narrow) pl.bn = 64;
    const int tiles = cdiv(P, pl.bm) * cdiv(Q, pl.bn);
    // split M so that the grid is ONE full wave of resident blocks: the TN kernels hold 3 blocks per CU (48 KB of LDS each), 768 on
    // the part.  Rounds 2-5 aimed at 1024 blocks -- a full wave plus a third of one, whose blocks run while two thirds of the slots
    // idle (round 6, TN_SLOTS: 1024 restores that)
#ifndef TN_SLOTS
#define TN_SLOTS 768
#endif
    int64_t want = TN_SLOTS / tiles;
    if (want < 1) want = 1;
    int64_t chunk = cdiv64(M, want);
    if (chunk < 256) chunk = 256;
    chunk = cdiv64(chunk, GEMM_BK) * GEMM_BK;
    pl.chunk = chunk;
    pl.splits = (int)cdiv64(M, chunk);
    return pl;
}

// ---- dense convolution on the GEMM kernels (called from dense.hip) ------------------------------------
static ConvGather make_gather(const ConvGemmGeom& g, bool dx_mode, const float* p0, const float* p1, int split,
                              const float* pfull = nullptr) {
    ConvGather cg;
    cg.pfull = pfull;
    if (!dx_mode) { cg.h = g.h; cg.w = g.w; cg.c = g.cin; cg.rh = g.ho; cg.rw = g.wo; }
    else { cg.h = g.ho; cg.w = g.wo; cg.c = g.cout; cg.rh = g.h; cg.rw = g.w; }
    cg.kw = g.kw; cg.sh = g.sh; cg.sw = g.sw; cg.ph = g.ph; cg.pw = g.pw; cg.dh = g.dh; cg.dw = g.dw;
    cg.p0 = p0; cg.p1 = p1; cg.split = split;
    cg.o_h = cg.o_w = cg.o_sy = cg.o_sx = cg.o_y0 = cg.o_x0 = 0;
    return cg;
}

bool conv_gemm_ok(const ConvGemmGeom& g) {   // vector gather (16-byte loads along the channels)
    return g.cin % 4 == 0 && g.cout % 4 == 0 && g.kh * g.kw * g.cin >= 32 && g.cout >= 16;
}
bool conv_gemm_elem_ok(const ConvGemmGeom& g) {   // element-wise gather: few input channels (stems), per-channel masks
    return g.cout % 4 == 0 && g.cout >= 16 && g.kh * g.kw * g.cin >= 24 && g.cin <= 16;
}

int launch_conv_gemm_fwd(const float* x, const float* mfull, RowScale rs, const float* wr, const float* bias,
                         const float* denom, const float* keep, const ConvGemmGeom& g, float* y, hipStream_t st, float* stats) {
    const int K = g.kh * g.kw * g.cin;
    const int64_t M = (int64_t)g.n * g.ho * g.wo;
    Epilogue ep = {denom, keep, bias, {nullptr, nullptr, 0}, 0, stats};
    const ConvGather cg = make_gather(g, false, rs.r0, rs.r1, rs.r0 != nullptr ? rs.split : 0, mfull);
    if (mfull != nullptr || g.cin % 4 != 0) return launch_nt_conv<3>(x, wr, K, y, g.cout, M, g.cout, K, ep, cg, st);
    return launch_nt_conv<1>(x, wr, K, y, g.cout, M, g.cout, K, ep, cg, st);
}

int launch_conv_gemm_dx(const float* dy, const float* inv, const float* wd, RowScale rs_out, const ConvGemmGeom& g,
                        float* dx, hipStream_t st) {
    const int K = g.kh * g.kw * g.cout;
    const int64_t M = (int64_t)g.n * g.h * g.w;
    Epilogue ep = {nullptr, nullptr, nullptr, rs_out, 0};
    const ConvGather cg = make_gather(g, true, inv, nullptr, 0x7fffffff);
    return launch_nt_conv<2>(dy, wd, K, dx, g.cin, M, g.cin, K, ep, cg, st);
}

// ---- dX of a strided conv (dilation 1) as sh*sw stride-1 problems --------------------------------------------
// Gathering all kh*kw taps for every input pixel and masking the non-divisible ones wastes (sh*sw - 1)/(sh*sw) of the
// MFMA work (measured: the 5x5 stride-2 dX of ImageFillOrigin took 9.4 ms against 2.5 ms for its forward).  Input pixels
// of phase (py, px) = (iy % sh, ix % sw) only ever see the taps ky = ky0 + sh*a, kx = kx0 + sw*b with
// ky0 = (py + ph) % sh: a stride-1 dX with the sub-sampled kernel, "padding" (py + ph - ky0) / sh, and its rows
// scattered back to the strided positions by the epilogue (ConvGather::o_*).
// w[co][ci][ky][kx] -> out[ci][((a*nb + b)*cout + co)]
__global__ void conv_w_layout_phase_kernel(const float* __restrict__ w, int cin, int cout, int kh, int kw, int ky0, int kx0,
                                           int sh, int sw, int na, int nb, float* __restrict__ out) {
    const int64_t total = (int64_t)cin * na * nb * cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout);
        const int t = (int)((i / cout) % (na * nb));
        const int64_t ci = i / ((int64_t)cout * na * nb);
        const int ky = ky0 + sh * (t / nb), kx = kx0 + sw * (t % nb);
        out[i] = w[(((int64_t)co * cin + ci) * kh + ky) * kw + kx];
    }
}

bool conv_gemm_dx_phases_ok(const ConvGemmGeom& g) {
    if (!(g.sh > 1 || g.sw > 1) || g.dh != 1 || g.dw != 1 || g.kh < g.sh || g.kw < g.sw) return false;
    const int min_taps = (g.kh / g.sh) * (g.kw / g.sw);          // fewest taps any phase sees
    return min_taps >= 1 && min_taps * g.cout >= 32;
}

// wd_ws: cin*cout*kh*kw floats (the phase layouts partition the taps)
int launch_conv_gemm_dx_phases(const float* dy, const float* inv, const float* w, float* wd_ws, RowScale rs_out,
                               const ConvGemmGeom& g, float* dx, hipStream_t st) {
    float* wp = wd_ws;
    for (int py = 0; py < g.sh; ++py)
        for (int px = 0; px < g.sw; ++px) {
            const int rh = (g.h - py + g.sh - 1) / g.sh, rw = (g.w - px + g.sw - 1) / g.sw;
            if (rh <= 0 || rw <= 0) continue;
            const int ky0 = (py + g.ph) % g.sh, kx0 = (px + g.pw) % g.sw;
            const int na = (g.kh - ky0 + g.sh - 1) / g.sh, nb = (g.kw - kx0 + g.sw - 1) / g.sw;
            const int K = na * nb * g.cout;
            hipLaunchKernelGGL(conv_w_layout_phase_kernel, dim3(stream_grid((int64_t)g.cin * K, 256)), dim3(256), 0, st, w, g.cin, g.cout,
                               g.kh, g.kw, ky0, kx0, g.sh, g.sw, na, nb, wp);
            int rc = check_launch("conv_w_layout_phase");
            if (rc) return rc;
            ConvGather cg;
            cg.pfull = nullptr;
            cg.h = g.ho; cg.w = g.wo; cg.c = g.cout; cg.rh = rh; cg.rw = rw;
            cg.kw = nb; cg.sh = 1; cg.sw = 1; cg.dh = 1; cg.dw = 1;
            cg.ph = (py + g.ph - ky0) / g.sh; cg.pw = (px + g.pw - kx0) / g.sw;
            cg.p0 = inv; cg.p1 = nullptr; cg.split = 0x7fffffff;
            cg.o_h = g.h; cg.o_w = g.w; cg.o_sy = g.sh; cg.o_sx = g.sw; cg.o_y0 = py; cg.o_x0 = px;
            Epilogue ep = {nullptr, nullptr, nullptr, rs_out, 0};
            const int64_t M = (int64_t)g.n * rh * rw;
            rc = launch_nt_conv<2>(dy, wp, K, dx, g.cin, M, g.cin, K, ep, cg, st);
            if (rc) return rc;
            wp += (size_t)g.cin * K;
        }
    return 0;
}

size_t conv_gemm_dw_ws_floats(const ConvGemmGeom& g) {
    const int K = g.kh * g.kw * g.cin;
    const TnPlan pl = plan_tn((int64_t)g.n * g.ho * g.wo, g.cout, K);
    return (size_t)pl.splits * g.cout * K;
}

int launch_conv_gemm_dw(const float* dy, const float* inv, const float* x, const float* mfull, RowScale rs,
                        const ConvGemmGeom& g, float* dwgt, float* ws, hipStream_t st) {
    const int K = g.kh * g.kw * g.cin, T = g.kh * g.kw;
    const int64_t M = (int64_t)g.n * g.ho * g.wo;
    const TnPlan pl = plan_tn(M, g.cout, K);
    const ConvGather cg = make_gather(g, false, rs.r0, rs.r1, rs.r0 != nullptr ? rs.split : 0, mfull);
    const RowScale none = {nullptr, nullptr, 0};
    dim3 grid(cdiv(K, pl.bn), cdiv(g.cout, pl.bm), pl.splits);
    const bool elem = (mfull != nullptr || g.cin % 4 != 0);
    // 64x64 split tiles lose to the f32 kernel on the narrow stem shapes (2M x 64 x 192: 1.43 vs 1.27 ms measured): the
    // gather form of the split kernel takes the 128x128 cases, and everything in the plain-bf16 mode
    if (!elem && (pl.big || gemm_products() == 1) && tn_split_conv_ok(dy, g.cout, x, M, g.cout, K, cg)) {
        int rc = launch_tn_split_conv(dy, g.cout, inv, x, cg, ws, M, g.cout, K, pl.chunk, pl.splits, pl.big, st);
        if (rc) return rc;
        return launch_reduce_rows_conv(ws, pl.splits, g.cout, g.cin, T, dwgt, st);
    }
    if (pl.big && !elem) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 2, true, 1>), grid, dim3(256), 0, st, dy, (int64_t)g.cout, inv, x, (int64_t)0, none, ws, M, g.cout, K, pl.chunk, cg, kNoBN);
    else if (!elem) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 1, 1, true, 1>), grid, dim3(256), 0, st, dy, (int64_t)g.cout, inv, x, (int64_t)0, none, ws, M, g.cout, K, pl.chunk, cg, kNoBN);
    else if (pl.big) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 2, true, 2>), grid, dim3(256), 0, st, dy, (int64_t)g.cout, inv, x, (int64_t)0, none, ws, M, g.cout, K, pl.chunk, cg, kNoBN);
    else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 1, 1, true, 2>), grid, dim3(256), 0, st, dy, (int64_t)g.cout, inv, x, (int64_t)0, none, ws, M, g.cout, K, pl.chunk, cg, kNoBN);
    int rc = check_launch("conv_gemm_tn");
    if (rc) return rc;
    return launch_reduce_rows_conv(ws, pl.splits, g.cout, g.cin, T, dwgt, st);
}

}  // namespace tsii

using namespace tsii;

extern "C" size_t tsii_pw_ws_bytes(int n, int k);

static int pw_fwd_impl(const float* x, int64_t m, int k, const float* w, int n, const float* bias, const float* r0, int split,
                       const float* r1, const float* denom, const float* keep, InBN ib, float* stats, float* y, void* ws, size_t ws_bytes,
                       void* stream, const float* up_add = nullptr, int up_h = 0, int up_w = 0) {
    TSII_REQUIRE(x && w && y, "pw_fwd: null pointer");
    TSII_REQUIRE(ws == nullptr || ws_bytes >= tsii_pw_ws_bytes(n, k), "pw_fwd: workspace too small");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0, "pw_fwd: bad shape m=%lld k=%d n=%d", (long long)m, k, n);
    RowScale as = {r0, r1, r0 != nullptr ? split : 0};
    Epilogue ep = {denom, keep, bias, {nullptr, nullptr, 0}, 0, stats};
    if (up_add != nullptr) {
        TSII_REQUIRE(up_h > 0 && up_w >= 4 && up_h % 2 == 0 && up_w % 4 == 0 && m % ((int64_t)up_h * up_w) == 0,
                     "pw_fwd_up: rows must be the pixels of whole images with even height and a width that is a multiple of 4 (got %d x %d, m=%lld)",
                     up_h, up_w, (long long)m);
        TSII_REQUIRE(n % 4 == 0 && aligned16(up_add) && m < (1ll << 31), "pw_fwd_up: n %% 4 == 0, a 16-byte aligned addend and m < 2^31 are required");
        ep.up_add = up_add;
        ep.up_w = (unsigned)up_w;
        make_up_div((unsigned)up_w, &ep.up_magic, &ep.up_shift);
    }
    return launch_nt(x, k, as, w, k, y, n, m, n, k, ep, (hipStream_t)stream, ib, ws);
}

extern "C" int tsii_pw_fwd(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                           const float* r0, int split, const float* r1, const float* denom, const float* keep,
                           float* y, void* ws, size_t ws_bytes, void* stream) {
    return pw_fwd_impl(x, m, k, w, n, bias, r0, split, r1, denom, keep, kNoBN, nullptr, y, ws, ws_bytes, stream);
}

extern "C" size_t tsii_pw_ws_bytes(int n, int k) {   // weight workspace of pw_fwd[_bn] (optional) and pw_bwd_dx[_bn] (wt_ws)
    if (n <= 0 || k <= 0) return 0;
    // forward: planes of W as [n][k]; dX: of W^T as [k][n] -- the tiled planes of the persistent kernel pad the REDUCTION dimension
    // to 32, so the two orientations differ whenever n or k is not a multiple of 32 (found on TextSegament: 48-channel layers)
    const size_t a = (size_t)n * k * sizeof(float), b = nt_split_ws_bytes(n, k), c = nt_pc_ws_bytes(n, k), d = nt_pc_ws_bytes(k, n);
    const size_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

extern "C" int64_t tsii_pw_stat_rows(int64_t m) { return m > 0 ? cdiv64(m, 128) : 0; }   // every NT tile variant has BM = 128

extern "C" int tsii_pw_fwd_bn(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                              const float* r0, int split, const float* r1, const float* denom, const float* keep,
                              const float* in_scale, const float* in_shift, int in_act, float in_slope,
                              float* stat_part, float* y, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pw_fwd_bn: in_scale / in_shift go together");
    InBN ib;
    TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "pw_fwd_bn: activation %d has no load-time form", in_act);
    return pw_fwd_impl(x, m, k, w, n, bias, r0, split, r1, denom, keep, ib, stat_part, y, ws, ws_bytes, stream);
}

extern "C" int tsii_pw_fwd_up(const float* x, int64_t m, int k, const float* w, int n, const float* bias,
                              const float* r0, int split, const float* r1, const float* denom, const float* keep,
                              const float* up_add, int up_h, int up_w, float* stat_part, float* y, void* ws, size_t ws_bytes,
                              void* stream) {
    TSII_REQUIRE(up_add != nullptr, "pw_fwd_up: null addend");
    return pw_fwd_impl(x, m, k, w, n, bias, r0, split, r1, denom, keep, kNoBN, stat_part, y, ws, ws_bytes, stream, up_add, up_h, up_w);
}

static int pw_bwd_dx_impl(const float* dy, int64_t m, int n, const float* w, int k, const float* inv,
                          const float* r0, int split, const float* r1, Epilogue ep, float* dx, float* wt_ws, void* stream) {
    TSII_REQUIRE(dy && w && dx && wt_ws, "pw_bwd_dx: null pointer");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0, "pw_bwd_dx: bad shape");
    RowScale as = {inv, nullptr, n};  // g = dy * inv[m]
    ep.cs = {r0, r1, split};
    // split-bf16 arithmetic: W^T is split into bf16 planes in wt_ws by one small kernel (transpose folded in)
    if (nt_split_ok(dy, n, w, k, n) && k % 4 == 0 && nt_pc_ok(dy, n, as, m, k, n, ep, kNoBN))
        return launch_nt_pc(dy, n, as, w, k, true, dx, k, m, k, n, ep, kNoBN, wt_ws, (hipStream_t)stream);
    if (nt_split_ok(dy, n, w, k, n) && k % 4 == 0)
        return launch_nt_split(dy, n, as, w, n, true, dx, k, m, k, n, ep, kNoBN, wt_ws, (hipStream_t)stream);
    // W [n,k] -> Wt [k,n] so both GEMM operands are contraction-contiguous
    int rc = launch_transpose(w, n, k, wt_ws, (hipStream_t)stream);
    if (rc) return rc;
    return launch_nt(dy, n, as, wt_ws, n, dx, k, m, k, n, ep, (hipStream_t)stream);
}

extern "C" int tsii_pw_bwd_dx(const float* dy, int64_t m, int n, const float* w, int k, const float* inv,
                              const float* r0, int split, const float* r1, float* dx, float* wt_ws, void* stream) {
    Epilogue ep = {};
    return pw_bwd_dx_impl(dy, m, n, w, k, inv, r0, split, r1, ep, dx, wt_ws, stream);
}

extern "C" int tsii_pw_bwd_dx_bn(const float* dy, int64_t m, int n, const float* w, int k, const float* inv,
                                 const float* r0, int split, const float* r1,
                                 const float* bn_y, const float* bn_mean, const float* bn_var, const float* bn_gamma,
                                 const float* bn_beta, float bn_eps, int bn_act, float bn_slope,
                                 float* dx, float* bwd_part, float* wt_ws, void* stream) {
    TSII_REQUIRE(bn_y && bn_mean && bn_var && bn_gamma && bn_beta && bwd_part, "pw_bwd_dx_bn: null pointer");
    InBN tmp;
    TSII_REQUIRE(make_in_bn(bn_mean, bn_var, bn_act, bn_slope, &tmp) == 0, "pw_bwd_dx_bn: activation %d has no load-time form", bn_act);
    Epilogue ep = {};
    ep.bn_y = bn_y; ep.bn_mean = bn_mean; ep.bn_var = bn_var; ep.bn_gamma = bn_gamma; ep.bn_beta = bn_beta;
    ep.bn_eps = bn_eps; ep.bn_neg = tmp.neg; ep.bn_hi = tmp.hi; ep.bn_part = bwd_part;
    return pw_bwd_dx_impl(dy, m, n, w, k, inv, r0, split, r1, ep, dx, wt_ws, stream);
}

extern "C" size_t tsii_pw_bwd_dw_ws_bytes(int64_t m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const TnPlan a = plan_tn(m, n, k, true), b = plan_tn(m, n, k, false), t = plan_tn(m, n, k, true, true);   // the call picks one by operand alignment / arithmetic mode
    int splits = a.splits > b.splits ? a.splits : b.splits;
    splits = t.splits > splits ? t.splits : splits;
    return ((size_t)splits * n * k + colsum_ws_floats(m, n)) * sizeof(float);
}

static int pw_bwd_dw_impl(const float* dy, const float* x, int64_t m, int n, int k, const float* inv, const float* keep,
                          const float* r0, int split, const float* r1, InBN ib, float* dw, float* dbias, void* ws,
                          size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && x && dw && ws, "pw_bwd_dw: null pointer");
    TSII_REQUIRE(m > 0 && k > 0 && n > 0, "pw_bwd_dw: bad shape");
    TSII_REQUIRE(ws_bytes >= tsii_pw_bwd_dw_ws_bytes(m, n, k), "pw_bwd_dw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (n % 4 == 0) && (k % 4 == 0) && aligned16(dy) && aligned16(x);
    const bool split_path = tn_split_ok(dy, n, x, k, n, k);
    TnPlan pl = plan_tn(m, n, k, vec, split_path);
    float* part = (float*)ws;
    RowScale sb = {r0, r1, split};
    dim3 grid(cdiv(k, pl.bn), cdiv(n, pl.bm), pl.splits);
    if (split_path) {     // split-bf16 MFMA (gemm_split.hip): same partial-slab workspace + row reduce
        int rc = launch_tn_split(dy, n, inv, x, k, sb, part, m, n, k, pl.chunk, pl.splits, pl.thin ? 3 : (pl.narrow ? 1 : (pl.big ? 0 : 2)), ib, st);
        if (rc) return rc;
        rc = launch_reduce_rows(part, pl.splits, (int64_t)n * k, dw, st);
        if (rc) return rc;
        if (dbias != nullptr) rc = launch_colsum_scaled(dy, keep, m, n, dbias, part + (size_t)pl.splits * n * k, st);
        return rc;
    }
    if (ib.sc != nullptr) {
        TSII_REQUIRE(vec, "pw_bwd_dw: input BatchNorm needs n, k %% 4 == 0 and 16-byte aligned operands");
        if (pl.narrow) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 1, true, 0, true>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, ib);
        else if (pl.big) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 2, true, 0, true>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, ib);
        else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 1, 1, true, 0, true>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, ib);
    } else if (pl.narrow) {
        hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 1, true, 0>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, kNoBN);
    } else if (pl.big) {
        if (vec) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 2, true, 0>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, kNoBN);
        else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 2, 2, false, 0>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, kNoBN);
    } else {
        if (vec) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 1, 1, true, 0>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, kNoBN);
        else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 1, 1, false, 0>), grid, dim3(256), 0, st, dy, (int64_t)n, inv, x, (int64_t)k, sb, part, m, n, k, pl.chunk, kNoConv, kNoBN);
    }
    int rc = check_launch("gemm_tn");
    if (rc) return rc;
    rc = launch_reduce_rows(part, pl.splits, (int64_t)n * k, dw, st);
    if (rc) return rc;
    if (dbias != nullptr) {
        float* cpart = part + (size_t)pl.splits * n * k;
        rc = launch_colsum_scaled(dy, keep, m, n, dbias, cpart, st);
    }
    return rc;
}

extern "C" int tsii_pw_bwd_dw(const float* dy, const float* x, int64_t m, int n, int k, const float* inv,
                              const float* keep, const float* r0, int split, const float* r1, float* dw, float* dbias,
                              void* ws, size_t ws_bytes, void* stream) {
    return pw_bwd_dw_impl(dy, x, m, n, k, inv, keep, r0, split, r1, kNoBN, dw, dbias, ws, ws_bytes, stream);
}

extern "C" int tsii_pw_bwd_dw_bn(const float* dy, const float* x, int64_t m, int n, int k, const float* inv,
                                 const float* keep, const float* r0, int split, const float* r1,
                                 const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                 float* dw, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(in_scale && in_shift, "pw_bwd_dw_bn: null scale / shift");
    InBN ib;
    TSII_REQUIRE(make_in_bn(in_scale, in_shift, in_act, in_slope, &ib) == 0, "pw_bwd_dw_bn: activation %d has no load-time form", in_act);
    return pw_bwd_dw_impl(dy, x, m, n, k, inv, keep, r0, split, r1, ib, dw, dbias, ws, ws_bytes, stream);
}
