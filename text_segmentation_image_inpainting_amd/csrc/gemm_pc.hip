// K3p: the split-bf16 NT GEMM (1x1 convolution forward / dX, gemm_split.hip) as a PERSISTENT PRODUCER / CONSUMER kernel.
//
// Why: in the 4-wave kernel of gemm_split.hip every wave does load -> split -> LDS -> barrier -> MFMA -> epilogue in
// turn; memory phases and matrix phases add up instead of overlapping (MFMA pipe 46 % busy, HBM 28 % at the same
// time).  Here the roles are separated by wave and a block walks many output tiles, so none of the phases ever
// drains:
//
//   768 threads = 12 waves = 3 per SIMD (a workgroup's waves go to the SIMDs round robin, so every SIMD hosts two
//   consumer waves and one producer wave), ONE block per CU, 168 VGPRs per lane.
//
//   consumers (waves 0-7):  wave (wm, wn) owns a 128-row x 32-column piece of the block tile (4 MFMA tiles, 64
//     accumulator registers).  Tile shapes: WM x WN = 1 x 8 -> 128 x 256, 2 x 4 -> 256 x 128.  Per 16-deep k step a
//     wave issues 24 v_mfma_f32_32x32x16_bf16 (6 partial products x 4 tiles) and, between them, the 15 ds_read_b128
//     of the NEXT k step's fragments (second register set for B, A registers recycled tile by tile), so after a
//     barrier the matrix pipe restarts from registers.  The epilogue runs straight from the accumulator layout
//     (lane = column): per-row factors come from a small LDS array the producers staged, BatchNorm statistics
//     (K6b) / BatchNorm-backward reductions (K6c) are per-lane sums over the 64 accumulators + one cross-half
//     shuffle, rows leave as 128-byte row segments (dword per lane).  While the consumers of a tile are in the
//     epilogue the producers are already two stages into the next tile.
//   producers (waves 8-11): stream the fp32 A rows (3 stages = 48 registers of loads in flight per lane at 128
//     rows), apply the producer's BatchNorm + activation (K6b) and the x*mask row scale, split into 3 bf16 planes
//     and write the XOR-swizzled LDS image of gemm_split.hip; B arrives pre-split and stage-tiled
//     ([k stage][plane][n][32 bf16], split_w_tiled_kernel) so a wave's 16-byte loads cover whole lines.
//
//   LDS: two stage buffers of 3 x (BM + BN) x 64 B = 72 KB.  One barrier per 16-deep k step: in interval j the
//   consumers multiply half-stage j from registers and read half-stage j + 1, the producers write stage
//   (j + 3) / 2 -- the buffer whose last reads completed before the previous barrier -- half of their items per
//   interval, and re-issue each item's global load for DA stages ahead right after its LDS store.
//   Hazards: a stage buffer is read in intervals 2s - 1 and 2s, rewritten (stage s + 2) in 2s + 1 and 2s + 2.
//
// Tiles are dealt to the blocks as contiguous ranges in (row block, column block) order, so the column blocks of one
// row block are consecutive on one CU (A re-reads hit L2) and every block streams B in the same order.
#include <stdlib.h>

#include "split_bf16.h"

namespace tsii {

struct PfYes { static constexpr bool value = true; };      // tags of the consumers' k step: prefetch the next fragments or not
struct PfNo { static constexpr bool value = false; };

struct PcCursor {      // a k stage of an output tile; wave-uniform
    unsigned tile;
    int ks;
    int64_t m0;
    int n0;
};

template <int BM, int BN>
__device__ __forceinline__ void pc_locate(PcCursor& c, unsigned tile, unsigned ntn) {
    c.tile = tile;
    c.m0 = (int64_t)(tile / ntn) * BM;
    c.n0 = (int)(tile % ntn) * BN;
}
// next stage; past the block's last tile the cursor stays on the last stage (loads issued from it are never used)
template <int BM, int BN>
__device__ __forceinline__ void pc_advance(PcCursor& c, int nst, unsigned ntn, unsigned tlast) {
    if (c.ks + 1 < nst) { ++c.ks; return; }
    if (c.tile < tlast) { c.ks = 0; pc_locate<BM, BN>(c, c.tile + 1, ntn); }
}

// epilogue of one consumer wave: 128 rows x 32 columns from acc[4] (D[row=(r&3)+8*(r>>2)+4*hi][col=li]).
// sideX / sideY: this tile's per-row factors in LDS (always staged, 1.0 when absent):
//   FWD:  y = keep ? acc * (1/denom) + bias : 0          (sideX = 1/denom, sideY = keep)
//   DX :  y = acc * (col < cs.split ? cs.r0 : cs.r1)      (sideX = cs.r0,  sideY = cs.r1)
// FULL: the wave's 128 x 32 piece lies inside the matrix (wave-uniform): no predicates on the stores / sums.
// Addresses: wave-uniform 64-bit bases + RUNNING 32-bit byte offsets (one add per row); written as 64 independent
// row * ldc products the compiler hoists all of them out of the tile loop and spills them.
template <bool BNB, bool FULL>
__device__ __forceinline__ void pc_epilogue_rows(f32x16 (&acc)[4], char* __restrict__ Cb, unsigned ldc4, const char* __restrict__ Yb, unsigned n4,
                                                 unsigned colb, int last, bool col_ok, int hi, const float* __restrict__ sideX,
                                                 const float* __restrict__ sideY, bool use_cs, bool cs_lo, float bias, bool stats, float pvt,
                                                 float bmu, float bis, float bga, float bbe, float bn_hi, float bn_neg, float& st1, float& st2) {
    unsigned coff = (unsigned)(4 * hi) * ldc4 + colb;
    unsigned yoff = (unsigned)(4 * hi) * n4 + colb;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);       // one 4-row band at a time: the unrolled epilogue must not pile up 64 rows of temporaries
            TSII_OPAQUE_U32(coff);                   // keep the offsets running (no re-derivation as row * ldc)
            TSII_OPAQUE_U32(yoff);
            const int rb4 = t * 32 + 8 * g + 4 * hi;
            float yv[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BNB) {     // raw BatchNorm input at the positions this lane stores (rows past the end read row 0 of the base)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = FULL || rb4 + j <= last;
                    yv[j] = *reinterpret_cast<const float*>(Yb + (ok ? yoff + (unsigned)j * n4 : colb));
                }
            }
            const float4 x4 = *reinterpret_cast<const float4*>(sideX + rb4);
            const float4 y4 = *reinterpret_cast<const float4*>(sideY + rb4);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float v;
                if (use_cs) v = acc[t][r] * (cs_lo ? xs[j] : ys[j]);
                else { v = fmaf(acc[t][r], xs[j], bias); v = ys[j] == 0.f ? 0.f : v; }
                const bool ok = FULL || (rb4 + j <= last && col_ok);
                if (stats) {
                    const float d = ok ? v - pvt : 0.f;
                    st1 += d;
                    st2 = fmaf(d, d, st2);
                }
                if constexpr (BNB) {
                    const float xh = (yv[j] - bmu) * bis;
                    const float z = fmaf(xh, bga, bbe);
                    float dz = v * ((z > 0.f && z < bn_hi) ? 1.f : (z > 0.f ? 0.f : bn_neg));
                    dz = ok ? dz : 0.f;
                    st1 += dz;
                    st2 = fmaf(dz, xh, st2);
                }
                if (ok) *reinterpret_cast<float*>(Cb + coff) = v;
                coff += ldc4;
            }
            coff += 4u * ldc4;
            yoff += 8u * n4;
        }
    }
}

template <bool BNB>
__device__ __forceinline__ void pc_epilogue(f32x16 (&acc)[4], float* __restrict__ C, int64_t ldc, int64_t M, int N, const Epilogue& ep,
                                            int64_t mw0, int col0, int li, int hi, const float* __restrict__ sideX,
                                            const float* __restrict__ sideY, bool use_cs) {
    const int col = col0 + li;
    const bool col_ok = col < N;
    const int colc = col_ok ? col : N - 1;
    const float bias = ep.bias != nullptr ? ep.bias[colc] : 0.f;
    const bool cs_lo = col < ep.cs.split;
    const int64_t left = M - mw0;                           // rows of this wave inside the matrix (may be <= 0 or > 128)
    const int last = (int)(left < 128 ? left : 128) - 1;    // last valid local row (< 0: none)
    float st1 = 0.f, st2 = 0.f, pvt = 0.f;
    const bool stats = ep.stats != nullptr;
    if (stats) {
        // pivot of the 128-row block: the value at its middle row (row 0 when the block is short), same for both lane halves
        const bool mid = left > 64;
        const float a0 = mid ? acc[2][0] : acc[0][0], x0 = mid ? sideX[64] : sideX[0], y0 = mid ? sideY[64] : sideY[0];
        float pv = fmaf(a0, x0, bias);
        pv = y0 == 0.f ? 0.f : pv;
        pvt = __shfl(pv, li, 64);
    }
    float bmu = 0.f, bis = 0.f, bga = 0.f, bbe = 0.f;
    if constexpr (BNB) {
        bmu = ep.bn_mean[colc]; bis = 1.0f / sqrtf(ep.bn_var[colc] + ep.bn_eps);
        bga = ep.bn_gamma[colc]; bbe = ep.bn_beta[colc];
    }
    char* __restrict__ Cb = reinterpret_cast<char*>(C + (last >= 0 ? mw0 : 0) * ldc);
    const char* __restrict__ Yb = BNB ? reinterpret_cast<const char*>(ep.bn_y + (last >= 0 ? mw0 : 0) * (int64_t)N) : nullptr;
    const unsigned ldc4 = (unsigned)ldc * 4u, n4 = (unsigned)N * 4u, colb = (unsigned)colc * 4u;
    if (last == 127 && col0 + 32 <= N)
        pc_epilogue_rows<BNB, true>(acc, Cb, ldc4, Yb, n4, colb, last, col_ok, hi, sideX, sideY, use_cs, cs_lo, bias, stats, pvt, bmu, bis, bga, bbe,
                                    ep.bn_hi, ep.bn_neg, st1, st2);
    else
        pc_epilogue_rows<BNB, false>(acc, Cb, ldc4, Yb, n4, colb, last, col_ok, hi, sideX, sideY, use_cs, cs_lo, bias, stats, pvt, bmu, bis, bga, bbe,
                                     ep.bn_hi, ep.bn_neg, st1, st2);
    if (stats || BNB) {
        st1 += __shfl_xor(st1, 32, 64);
        st2 += __shfl_xor(st2, 32, 64);
        if (hi == 0 && col_ok && left > 0) {
            const int64_t rb = mw0 >> 7;                    // 128-row block index (tsii_pw_stat_rows)
            if constexpr (BNB) {
                float* sp = ep.bn_part + rb * 2 * N;
                sp[col] = st1;
                sp[N + col] = st2;
            } else {
                float* sp = ep.stats + rb * 4 * N;
                sp[col] = (float)(left < 128 ? left : 128);
                sp[N + col] = pvt;
                sp[2 * N + col] = st1;
                sp[3 * N + col] = st2;
            }
        }
    }
}

template <int WM, int WN, int PRODUCTS, bool BNIN, bool BNB>
__global__ __launch_bounds__(768, 3) void gemm_nt_pc_kernel(const float* __restrict__ A, int64_t lda, RowScale as,
                                                            const unsigned short* __restrict__ Bp, float* __restrict__ C, int64_t ldc,
                                                            int64_t M, int N, int K, Epilogue ep, InBN ib, unsigned ntn, unsigned tiles) {
    static_assert(WM * WN == 8, "8 consumer waves");
    constexpr int P = SplitPlanes<PRODUCTS>::value;
    constexpr int BM = WM * 128, BN = WN * 32;
    constexpr int STAGE = P * (BM + BN) * 64;                 // bytes of one stage buffer
    constexpr int DA = (BM == 128) ? 3 : 2;                   // stages of A loads in flight per producer lane
    constexpr int NA = BM / 64;                               // A items (row, 8-k chunk) per producer thread and stage
    constexpr int NBP = BN / 64;                              // B pieces per plane, producer thread and stage
    constexpr int SIDE_FLOATS = 2 * 2 * BM;                   // [tile parity][X, Y][row]
    constexpr int BNV_FLOATS = BNIN ? 2048 : 0;               // input BatchNorm (scale, shift) of all K <= 1024 channels
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + (SIDE_FLOATS + BNV_FLOATS) * 4];
    float* side = reinterpret_cast<float*>(smem + 2 * STAGE);
    float* bnv = side + SIDE_FLOATS;

    const int tid = threadIdx.x;
    // this block's tiles: [t0, t1)
    const unsigned t0 = (unsigned)(((uint64_t)blockIdx.x * tiles) / gridDim.x);
    const unsigned t1 = (unsigned)(((uint64_t)(blockIdx.x + 1) * tiles) / gridDim.x);
    const int nst = (K + 31) >> 5;
    const unsigned stages = (t1 - t0) * (unsigned)nst;        // >= 1: the launcher never starts more blocks than tiles
    // epilogue modes (wave-uniform): FWD (1/denom, keep) or DX (cs.r0, cs.r1); the launcher rejects both at once
    const bool use_cs = ep.cs.r0 != nullptr;

    if constexpr (BNIN) {
        for (int i = tid; i < 2048; i += 768) {
            const int k = i & 1023;
            bnv[i] = k < K ? (i < 1024 ? ib.sc[k] : ib.sh[k]) : 0.f;
        }
        __syncthreads();
    }

    if (tid < 512) {
        // ------------------------------------------------ consumers ------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        const int cw = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
        const int wm = cw / WN, wn = cw % WN;
        const int swz = (li >> 2) & 3;
        const int aoff = (wm * 128 + li) * 64;                          // + t * 2048 + p * BM * 64
        const int boff = P * BM * 64 + (wn * 32 + li) * 64;             // + p * BN * 64
        const int ch0 = ((0 + hi) ^ swz) << 4, ch1 = ((2 + hi) ^ swz) << 4;   // chunk of k half 0 / 1

        bf16x8 a[4][P], b[P];
        f32x16 acc[4];
        auto ldf = [](const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); };
        // One 16-deep k step from registers, hand-scheduled (sched_barrier pins the groups): the fragments of the NEXT step
        // (stage buffer Sn, chunk chn) are read between the MFMA groups into registers that are free by then -- B and the
        // last A tile into a second set, the other A tiles into their own registers once their products are issued -- so
        // only 3 reads trail the last 3 MFMAs and the step after the barrier starts from registers.
        auto kstep = [&](const unsigned char* Sn, int chn, auto pf_tag) {
            constexpr bool PF = decltype(pf_tag)::value;              // false: the last step of a tile (its fragments die in the epilogue)
            bf16x8 bn[P], a3n[P];
            if constexpr (PF) {
#pragma unroll
                for (int p = 0; p < P; ++p) bn[p] = ldf(Sn + boff + p * (BN * 64) + chn);
#pragma unroll
                for (int p = 0; p < P; ++p) a3n[p] = ldf(Sn + aoff + 3 * 2048 + p * (BM * 64) + chn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PRODUCTS; ++q) {
                const int pa = SplitTerm<PRODUCTS>::pa(q), pb = SplitTerm<PRODUCTS>::pb(q);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][pa], b[pb], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][pa], b[pb], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < P; ++p) a[t][p] = ldf(Sn + aoff + t * 2048 + p * (BM * 64) + chn);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int QH = PRODUCTS / 2;
#pragma unroll
            for (int q = 0; q < QH; ++q) {
                const int pa = SplitTerm<PRODUCTS>::pa(q), pb = SplitTerm<PRODUCTS>::pb(q);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][pa], b[pb], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3][pa], b[pb], acc[3], 0, 0, 0);
            }
#pragma unroll
            for (int q = QH; q < PRODUCTS; ++q)
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][SplitTerm<PRODUCTS>::pa(q)], b[SplitTerm<PRODUCTS>::pb(q)], acc[2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {
#pragma unroll
                for (int p = 0; p < P; ++p) a[2][p] = ldf(Sn + aoff + 2 * 2048 + p * (BM * 64) + chn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = QH; q < PRODUCTS; ++q)
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3][SplitTerm<PRODUCTS>::pa(q)], b[SplitTerm<PRODUCTS>::pb(q)], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {
#pragma unroll
                for (int p = 0; p < P; ++p) { b[p] = bn[p]; a[3][p] = a3n[p]; }
            }
        };
        auto load_frags = [&](const unsigned char* Sn, int chn) {
#pragma unroll
            for (int p = 0; p < P; ++p) b[p] = ldf(Sn + boff + p * (BN * 64) + chn);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < P; ++p) a[t][p] = ldf(Sn + aoff + t * 2048 + p * (BM * 64) + chn);
        };

        __syncthreads();                                                // stage 0 and the first half of stage 1 are in LDS

        // Per tile: fragments of its first k step (complete since the last barrier), the k loop, the epilogue.  Nothing but
        // the accumulators lives across the epilogue, and the fragment registers are carried by the inner loop only (with
        // the prefetch carried across tiles the allocator spilled fragments inside the k loop).
        unsigned g = 0;                                                 // stage counter of the block
        for (unsigned tile = t0; tile < t1; ++tile) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            load_frags(smem + (g & 1u) * STAGE, ch0);
            for (int ks = 0; ks < nst; ++ks, ++g) {
                const unsigned char* S0 = smem + (g & 1u) * STAGE;
                const unsigned char* S1 = smem + ((g + 1u) & 1u) * STAGE;
                kstep(S0, ch1, PfYes());                                // k half 0 of stage g; fetch its half 1
                __syncthreads();
                kstep(S1, ch0, PfYes());                                // k half 1; fetch half 0 of stage g + 1 (unused after the tile's last stage)
                __syncthreads();
            }
            const int64_t m0 = (int64_t)(tile / ntn) * BM;
            const int n0 = (int)(tile % ntn) * BN;
            const float* sx = side + (tile & 1u) * (2 * BM) + wm * 128;
            pc_epilogue<BNB>(acc, C, ldc, M, N, ep, m0 + wm * 128, n0 + wn * 32, li, hi, sx, sx + BM, use_cs);
        }
    } else {
        // ------------------------------------------------ producers ------------------------------------------------
        const int ptid = tid - 512;
        const int prow = ptid >> 2, pch = ptid & 3;                     // item i: row prow + 64 i of the tile, chunk pch
        const unsigned tlast = t1 - 1;
        PcCursor la, lb, wc;                                            // A loads, B loads, LDS writes
        pc_locate<BM, BN>(la, t0, ntn); la.ks = 0;
        lb = la; wc = la;

        float4 ra[DA][NA][2];
        float sa0[DA][NA], sa1[DA][NA];
        u32x4 rb[P * NBP];
        const bool has_r0 = as.r0 != nullptr, has_r1 = as.r1 != nullptr;
        const float* r0p = has_r0 ? as.r0 : A;                          // branch-free: a dummy (valid) address when absent
        const float* r1p = has_r1 ? as.r1 : r0p;

        auto load_a = [&](const PcCursor& c, int slot, int i) {
            const int64_t rowl = c.m0 + prow + 64 * i;
            const int64_t row = rowl < M ? rowl : M - 1;
            int k = c.ks * 32 + pch * 8;
            k = k < K - 8 ? k : K - 8;                                  // K % 8 == 0: the stage-tiled B holds zeros past K
            const float* p = A + row * lda + k;
            ra[slot][i][0] = *reinterpret_cast<const float4*>(p);
            ra[slot][i][1] = *reinterpret_cast<const float4*>(p + 4);
            const float x0 = r0p[has_r0 ? row : 0], x1 = r1p[(has_r0 || has_r1) ? row : 0];
            sa0[slot][i] = has_r0 ? x0 : 1.f;
            sa1[slot][i] = has_r1 ? x1 : 1.f;
        };
        auto load_b = [&](const PcCursor& c, int e) {                   // piece e = p * NBP + ii
            const int p = e / NBP, ii = e % NBP;
            int nrow = c.n0 + prow + 64 * ii;
            nrow = nrow < N ? nrow : N - 1;
            const unsigned off = (unsigned)((((c.ks * P + p) * N + nrow) * 32 + pch * 8) * 2);
            rb[e] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Bp) + off);
        };
        auto store_a = [&](const PcCursor& c, unsigned char* S, int slot, int i) {
            float v[8] = {ra[slot][i][0].x, ra[slot][i][0].y, ra[slot][i][0].z, ra[slot][i][0].w,
                          ra[slot][i][1].x, ra[slot][i][1].y, ra[slot][i][1].z, ra[slot][i][1].w};
            int k = c.ks * 32 + pch * 8;
            k = k < K - 8 ? k : K - 8;
            if constexpr (BNIN) {
                const float4 c0 = *reinterpret_cast<const float4*>(bnv + k), c1 = *reinterpret_cast<const float4*>(bnv + k + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(bnv + 1024 + k), h1 = *reinterpret_cast<const float4*>(bnv + 1024 + k + 4);
                const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bn_act_load(v[e], sc[e], sh[e], ib.neg, ib.hi);
            }
            if (has_r0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (k + e < as.split) ? sa0[slot][i] : sa1[slot][i];
            }
            u32x4 pl[P];
            split8<P>(v, pl);
            const int off = split_off(prow + 64 * i, pch);
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (BM * 64) + off) = pl[p];
        };
        auto store_b = [&](unsigned char* S, int e) {
            const int p = e / NBP, ii = e % NBP;
            *reinterpret_cast<u32x4*>(S + P * BM * 64 + p * (BN * 64) + split_off(prow + 64 * ii, pch)) = rb[e];
        };
        // per-row epilogue factors of the tile at c (c.ks == 0): loads first, LDS stores when the interval's other work is done
        float sdx = 1.f, sdy = 1.f;
        int srow = 0;
        auto side_load = [&](const PcCursor& c, int rr) {
            const int64_t rowl = c.m0 + rr;
            const int64_t row = rowl < M ? rowl : M - 1;
            sdx = 1.f; sdy = 1.f;
            if (use_cs) { sdx = ep.cs.r0[row]; if (ep.cs.r1 != nullptr) sdy = ep.cs.r1[row]; }
            else {
                if (ep.denom != nullptr) sdx = ep.denom[row];
                if (ep.keep != nullptr) sdy = ep.keep[row];
            }
            srow = rr;
        };
        auto side_store = [&](const PcCursor& c) {
            float* sx = side + (c.tile & 1u) * (2 * BM);
            sx[srow] = (!use_cs && ep.denom != nullptr) ? 1.0f / sdx : sdx;     // one IEEE division per row
            sx[BM + srow] = sdy;
        };

        // ---- prologue: loads of stages 0 .. DA-1 (A) and 0 (B); stage 0 and the first half of stage 1 into LDS ----
#pragma unroll
        for (int u = 0; u < DA; ++u) {
#pragma unroll
            for (int i = 0; i < NA; ++i) load_a(la, u, i);
            pc_advance<BM, BN>(la, nst, ntn, tlast);
        }
#pragma unroll
        for (int e = 0; e < P * NBP; ++e) load_b(lb, e);
        pc_advance<BM, BN>(lb, nst, ntn, tlast);
        for (int rr = ptid; rr < BM; rr += 256) { side_load(wc, rr); side_store(wc); }
#pragma unroll
        for (int i = 0; i < NA; ++i) { store_a(wc, smem, 0, i); load_a(la, 0, i); }
        pc_advance<BM, BN>(la, nst, ntn, tlast);
#pragma unroll
        for (int e = 0; e < P * NBP; ++e) { store_b(smem, e); load_b(lb, e); }
        pc_advance<BM, BN>(lb, nst, ntn, tlast);
        pc_advance<BM, BN>(wc, nst, ntn, tlast);                        // wc = stage 1
        if (stages > 1) {
#pragma unroll
            for (int i = 0; i < NA / 2; ++i) { store_a(wc, smem + STAGE, 1 % DA, i); load_a(la, 1 % DA, i); }
#pragma unroll
            for (int e = 0; e < P * NBP / 2; ++e) { store_b(smem + STAGE, e); load_b(lb, e); }
        }
        __syncthreads();

        // ---- main loop: interval 2s-2 = second half of stage s, interval 2s-1 = first half of stage s+1 ----
        for (unsigned sb = 1; sb <= stages; sb += DA) {
#pragma unroll
            for (int u = 0; u < DA; ++u) {
                const unsigned s = sb + u;
                if (s > stages) break;
                const int slot = (1 + u) % DA, slot1 = (2 + u) % DA;
                if (s < stages) {
                    unsigned char* S = smem + (s & 1u) * STAGE;
                    const bool newtile = wc.ks == 0;
                    if (newtile) side_load(wc, ptid);                   // (BM == 128: the upper half loads clamped rows it never stores)
#pragma unroll
                    for (int i = NA / 2; i < NA; ++i) { store_a(wc, S, slot, i); load_a(la, slot, i); }
                    pc_advance<BM, BN>(la, nst, ntn, tlast);
#pragma unroll
                    for (int e = P * NBP / 2; e < P * NBP; ++e) { store_b(S, e); load_b(lb, e); }
                    pc_advance<BM, BN>(lb, nst, ntn, tlast);
                    if (newtile && ptid < BM) side_store(wc);
                    pc_advance<BM, BN>(wc, nst, ntn, tlast);
                }
                __syncthreads();
                if (s + 1 < stages) {
                    unsigned char* S = smem + ((s + 1u) & 1u) * STAGE;
#pragma unroll
                    for (int i = 0; i < NA / 2; ++i) { store_a(wc, S, slot1, i); load_a(la, slot1, i); }
#pragma unroll
                    for (int e = 0; e < P * NBP / 2; ++e) { store_b(S, e); load_b(lb, e); }
                }
                __syncthreads();
            }
        }
    }
}

// ---- weights: fp32 [N,K] (or its transpose) -> P bf16 planes in the stage-tiled layout [k stage][plane][n][32] ------
template <int P>
__global__ void split_w_tiled_kernel(const float* __restrict__ w, int cols_in, int transpose, int N, int K, int nst,
                                     unsigned short* __restrict__ planes) {
    const int64_t total = (int64_t)nst * N * 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 31);
        const int n = (int)((i >> 5) % N), s = (int)((i >> 5) / N);
        const int k = s * 32 + kk;
        float x = 0.f;
        if (k < K) x = transpose ? w[(int64_t)k * cols_in + n] : w[(int64_t)n * cols_in + k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const __bf16 h = (__bf16)x;                               // RNE
            const unsigned short u = __builtin_bit_cast(unsigned short, h);
            planes[(((int64_t)s * P + p) * N + n) * 32 + kk] = u;
            x -= __builtin_bit_cast(float, (unsigned)u << 16);
        }
    }
}

static int pc_cus() {            // read-only device-properties cache
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus = v;
    }
    return cus;
}

static int g_pc = getenv("TSII_GEMM_PC") ? atoi(getenv("TSII_GEMM_PC")) : 1;              // A/B knob: 0 = 4-wave kernels only
static int g_pc_min_n = getenv("TSII_GEMM_PC_MIN_N") ? atoi(getenv("TSII_GEMM_PC_MIN_N")) : 64;

size_t nt_pc_ws_bytes(int n, int k) { return (size_t)3 * n * ((k + 31) & ~31) * sizeof(unsigned short) + 16; }

bool nt_pc_ok(const float* A, int64_t lda, int N, int K, const Epilogue& ep, const InBN& ib) {
    if (!g_pc || gemm_products() != 6) return false;
    if (N < g_pc_min_n || K % 8 != 0 || lda % 4 != 0 || !aligned16(A)) return false;
    if (ep.cs.r0 != nullptr && (ep.denom != nullptr || ep.keep != nullptr || ep.bias != nullptr)) return false;   // one epilogue mode at a time
    if (ib.sc != nullptr && K > 1024) return false;                                                               // (scale, shift) live in LDS
    if ((int64_t)3 * N * ((K + 31) & ~31) * 2 >= (1ll << 31)) return false;
    return true;
}

template <int WM, int WN>
static int launch_nt_pc_cfg(const float* A, int64_t lda, RowScale as, const unsigned short* Bp, float* C, int64_t ldc,
                            int64_t M, int N, int K, Epilogue ep, InBN ib, hipStream_t stream) {
    constexpr int BM = WM * 128, BN = WN * 32;
    const unsigned ntn = (unsigned)cdiv(N, BN);
    const int64_t tiles = cdiv64(M, BM) * ntn;
    TSII_REQUIRE(tiles < (1ll << 31), "gemm_nt_pc: too many tiles");
    const unsigned grid = (unsigned)(tiles < pc_cus() ? tiles : pc_cus());
    if (ep.bn_y != nullptr) {
        TSII_REQUIRE(ib.sc == nullptr, "gemm_nt_pc: no input BatchNorm together with the BatchNorm-backward epilogue");
        hipLaunchKernelGGL((gemm_nt_pc_kernel<WM, WN, 6, false, true>), dim3(grid), dim3(768), 0, stream, A, lda, as, Bp, C, ldc, M, N, K, ep, ib, ntn, (unsigned)tiles);
    } else if (ib.sc != nullptr) {
        hipLaunchKernelGGL((gemm_nt_pc_kernel<WM, WN, 6, true, false>), dim3(grid), dim3(768), 0, stream, A, lda, as, Bp, C, ldc, M, N, K, ep, ib, ntn, (unsigned)tiles);
    } else {
        hipLaunchKernelGGL((gemm_nt_pc_kernel<WM, WN, 6, false, false>), dim3(grid), dim3(768), 0, stream, A, lda, as, Bp, C, ldc, M, N, K, ep, ib, ntn, (unsigned)tiles);
    }
    return check_launch("gemm_nt_pc");
}

// B = fp32 [N,K] (b_transposed: fp32 [K,N]); wsplit: nt_pc_ws_bytes(N, K) bytes
int launch_nt_pc(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, bool b_transposed, float* C, int64_t ldc,
                 int64_t M, int N, int K, Epilogue ep, InBN ib, void* wsplit, hipStream_t stream) {
    unsigned short* planes = reinterpret_cast<unsigned short*>((reinterpret_cast<uintptr_t>(wsplit) + 15) & ~(uintptr_t)15);
    const int nst = (K + 31) >> 5;
    hipLaunchKernelGGL(split_w_tiled_kernel<3>, dim3(stream_grid((int64_t)nst * N * 32, 256)), dim3(256), 0, stream, B, (int)ldb, b_transposed ? 1 : 0, N, K, nst, planes);
    int rc = check_launch("split_w_tiled");
    if (rc) return rc;
    // 256-column tiles when they waste no more columns than 128-column ones
    if (cdiv(N, 256) * 256 == cdiv(N, 128) * 128) return launch_nt_pc_cfg<1, 8>(A, lda, as, planes, C, ldc, M, N, K, ep, ib, stream);
    return launch_nt_pc_cfg<2, 4>(A, lda, as, planes, C, ldc, M, N, K, ep, ib, stream);
}

}  // namespace tsii
