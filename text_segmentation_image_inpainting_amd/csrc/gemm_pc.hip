// K3p: the split-bf16 NT GEMM (1x1 convolution forward / dX, gemm_split.hip) as a PERSISTENT PRODUCER / CONSUMER kernel.
//
// Why: in the 4-wave kernel of gemm_split.hip every wave does load -> split -> LDS -> barrier -> MFMA -> epilogue in
// turn; memory phases and matrix phases add up instead of overlapping (MFMA pipe 46 % busy, HBM 28 % at the same
// time).  Here the roles are separated by wave, a block walks many output tiles, and NO block-wide barrier exists after
// the start: waves meet through per-stage LDS counters only, so a wave waits for data, never for its neighbours.
//
//   768 threads = 12 waves = 3 per SIMD (a workgroup's waves go to the SIMDs round robin: every SIMD hosts two consumer
//   waves and one producer wave), ONE block per CU, <= 168 VGPRs per lane.
//
//   producers (waves 8-11) own the fp32 operand A: rows are requested DA stages ahead (5 x 16 KB per CU at 128-row
//     tiles) with counted asynchronous loads (tsii_common.h: hipcc's own waitcnt insertion made this loop run at one
//     memory latency per stage), get the producer layer's BatchNorm + activation (K6b) and the x*mask row scale, are
//     split into 3 bf16 planes and written into a ring of R LDS stages ([plane][row][32 k], 64-byte rows, 16-byte
//     chunks XOR-swizzled as in gemm_split.hip).  Per stage a wave waits for empty[slot] (all 8 consumers done with
//     the slot's previous use) and bumps full[slot] after its stores.
//   consumers (waves 0-7): wave (wm, wn) owns 128 rows x 32 columns of the block tile (4 MFMA tiles, 64 accumulators;
//     WM x WN = 1 x 8 -> 128 x 256 tiles, 2 x 4 -> 256 x 128).  Its B fragments never touch LDS: the weights arrive
//     pre-split and tiled as [k half-step][plane][n][16 k] (split_w_tiled_kernel), i.e. exactly one MFMA operand row
//     (32 bytes) per output column, and every lane loads its 16 bytes straight from L2 one k step ahead.  Per 16-deep k
//     step a wave issues 24 v_mfma_f32_32x32x16_bf16 (6 partial products x 4 tiles) and, between them, the next
//     step's 12 ds_read_b128 (A) + 3 global loads (B) into registers that are free by then (hand-scheduled,
//     sched_barrier).  It polls full[slot] before the first fragment read of a stage and releases empty[slot] when its
//     last fragment reads of the stage have been issued (the release orders them).
//   Epilogue straight from the accumulator layout (lane = column): per-row factors from a small LDS ring the
//     producers fill (1/denominator and keep for the forward, the two mask planes for dX), BatchNorm statistics (K6b)
//     / BatchNorm-backward reductions (K6c) as per-lane sums + one cross-half shuffle, rows leave as 128-byte row
//     segments.  The producers are up to R stages into the next tiles meanwhile.
//
// Tiles are dealt to the blocks as contiguous ranges in (row block, column block) order, so the column blocks of one
// row block are consecutive on one CU (A re-reads hit L2) and every block streams B in the same order.
#include <stdlib.h>

#include "split_bf16.h"

#ifndef PC_NT_STORE
#define PC_NT_STORE 1     // non-temporal stores of the output tile (A/B on the chip: forward 17.88 -> 17.62 ms)
#endif
#ifndef PC_WIDE_K
#define PC_WIDE_K 0       // A/B: reductions up to this length run 128 x 256 tiles for every N (fewer column tiles = fewer passes over A)
#endif
#ifndef PC_OPT_DEFAULT
#define PC_OPT_DEFAULT 16 // the kernel's `opt` word in the stock library: bit 4 = tiles dealt round robin per XCD, bits 5-6 = consumer lag
#endif

namespace tsii {

struct PcCursor {      // a k stage of an output tile; wave-uniform
    unsigned tile;     // global tile index
    unsigned ord;      // its position in this block's sequence of tiles
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
// next stage; past the block's last tile the cursor stays on the last stage (loads issued from it are never used).  A block's tiles
// are tfirst + i * tstride, i < tcount.
template <int BM, int BN>
__device__ __forceinline__ void pc_advance(PcCursor& c, int nst, unsigned ntn, unsigned tcount, unsigned tstride) {
    if (c.ks + 1 < nst) { ++c.ks; return; }
    if (c.ord + 1 < tcount) { c.ks = 0; ++c.ord; pc_locate<BM, BN>(c, c.tile + tstride, ntn); }
}

__device__ __forceinline__ unsigned pc_flag(const unsigned* f) { return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pc_wait_flag(const unsigned* f, unsigned need) {
    while (pc_flag(f) < need) __builtin_amdgcn_s_sleep(1);
}
// the producers' form: they run R stages ahead, so a slow poll costs nothing -- and every instruction a spinning wave issues
// is taken from the matrix stream of its SIMD
__device__ __forceinline__ void pc_wait_flag_lazy(const unsigned* f, unsigned need) {
    while (pc_flag(f) < need) __builtin_amdgcn_s_sleep(6);
}
__device__ __forceinline__ void pc_bump_flag(unsigned* f) { __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Epilogue of one consumer wave: 128 rows x 32 columns from acc[4] (D[row=(r&3)+8*(r>>2)+4*hi][col=li]); the launcher
// only sends shapes whose tiles are full in M and whose columns come in whole 32-blocks (a wave past N skips it).
// sideX / sideY: the tile's per-row factors in LDS (always staged, 1.0 when absent).  EPI:
//   0 / 1 (forward without / with BatchNorm statistics):  y = keep ? acc * (1/denom) + bias : 0      (sideX = 1/denom, sideY = keep)
//   2 / 3 (dX without / with the K6c reductions):          y = acc * (col < cs.split ? cs.r0 : cs.r1)  (sideX = cs.r0,  sideY = cs.r1)
//   4 / 5 (forward 0 / 1 with the up-sampled addend, gemm_tiles.h: Epilogue::up_add):  acc + up_add[low row][col] in place of acc;
//          a 4-row band (rows 4-aligned, up_w % 4 == 0) lies in one image row, so it reads TWO addend rows, one band ahead
// Addresses: wave-uniform 64-bit bases + RUNNING 32-bit byte offsets (one add per row); written as 64 independent
// row * ldc products the compiler hoists all of them out of the tile loop and spills them.
template <int EPI>
__device__ __forceinline__ void pc_epilogue(f32x16 (&acc)[4], float* __restrict__ C, int64_t ldc, int N, const Epilogue& ep,
                                            int64_t mw0, int col0, int li, int hi, const float* __restrict__ sideX,
                                            const float* __restrict__ sideY, float bias) {
    constexpr bool UP = EPI >= 4, DX = EPI == 2 || EPI == 3, STATS = EPI == 1 || EPI == 5, BNB = EPI == 3;
    const int col = col0 + li;
    const bool cs_lo = col < ep.cs.split;
    float st1 = 0.f, st2 = 0.f, pvt = 0.f;
    const char* __restrict__ Zb = UP ? reinterpret_cast<const char*>(ep.up_add) : nullptr;
    const unsigned mw0u = (unsigned)mw0;
    auto zoff_of = [&](int rb4) {       // byte offset of the addend row of band rows rb4, rb4 + 1 (rows rb4 + 2, + 3: the next addend row)
        return up_low_row(mw0u + (unsigned)rb4, ep.up_w, ep.up_magic, ep.up_shift) * ((unsigned)N * 4u) + (unsigned)col * 4u;
    };
    if constexpr (STATS) {      // pivot of the 128-row block: the value at its middle row, same for both lane halves
        float pv = fmaf(acc[2][0], sideX[64], bias);
        pv = sideY[64] == 0.f ? 0.f : pv;
        pvt = __shfl(pv, li, 64);
    }
    float bmu = 0.f, bis = 0.f, bga = 0.f, bbe = 0.f;
    if constexpr (BNB) {
        bmu = ep.bn_mean[col]; bis = 1.0f / sqrtf(ep.bn_var[col] + ep.bn_eps);
        bga = ep.bn_gamma[col]; bbe = ep.bn_beta[col];
    }
    char* __restrict__ Cb = reinterpret_cast<char*>(C + mw0 * ldc);
    const char* __restrict__ Yb = BNB ? reinterpret_cast<const char*>(ep.bn_y + mw0 * (int64_t)N) : nullptr;
    const unsigned ldc4 = (unsigned)ldc * 4u, n4 = (unsigned)N * 4u;
    unsigned coff = (unsigned)(4 * hi) * ldc4 + (unsigned)col * 4u;
    unsigned yoff = (unsigned)(4 * hi) * n4 + (unsigned)col * 4u;
    // the row factors of a band are fetched from LDS one band ahead (read where they are used, each band started with a full LDS
    // round trip: 16 of them per tile and wave)
    float4 xq = *reinterpret_cast<const float4*>(sideX + 4 * hi), yq = *reinterpret_cast<const float4*>(sideY + 4 * hi);
    float ynext[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BNB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ynext[j] = *reinterpret_cast<const float*>(Yb + yoff + (unsigned)j * n4);
    }
    // all 16 bands' addend rows are requested up front (32 registers: the A / B fragment registers are dead here): one memory round
    // trip per tile -- a band ahead left these HBM-bound K = 64 / 128 layers waiting on every band (1.33 ms for 2M x 64 -> 384)
    float zall[UP ? 16 : 1][2];
    if constexpr (UP) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const unsigned zo = zoff_of((b >> 2) * 32 + 8 * (b & 3) + 4 * hi);
            zall[b][0] = *reinterpret_cast<const float*>(Zb + zo);
            zall[b][1] = *reinterpret_cast<const float*>(Zb + zo + n4);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);       // one 4-row band at a time: the unrolled epilogue must not pile up 64 rows of temporaries
            TSII_OPAQUE_U32(coff);                   // keep the offsets running (no re-derivation as row * ldc)
            TSII_OPAQUE_U32(yoff);
            const int rb4 = t * 32 + 8 * g + 4 * hi;
            float yv[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BNB) {     // raw BatchNorm input at the positions this lane stores, requested one band ahead (these
                                     // layers are HBM-bound: 4 loads in flight per wave left the epilogue waiting on each band)
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[j] = ynext[j];
                if (t * 4 + g < 15) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ynext[j] = *reinterpret_cast<const float*>(Yb + yoff + 8u * n4 + (unsigned)j * n4);
                }
            }
            float zv[2] = {0.f, 0.f};
            if constexpr (UP) { zv[0] = zall[t * 4 + g][0]; zv[1] = zall[t * 4 + g][1]; }
            const float4 x4 = xq, y4 = yq;
            if (t * 4 + g < 15) {
                xq = *reinterpret_cast<const float4*>(sideX + rb4 + 8);
                yq = *reinterpret_cast<const float4*>(sideY + rb4 + 8);
            }
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float v;
                if constexpr (DX) v = acc[t][r] * (cs_lo ? xs[j] : ys[j]);
                else if constexpr (UP) { v = fmaf(acc[t][r] + zv[j >> 1], xs[j], bias); v = ys[j] == 0.f ? 0.f : v; }
                else { v = fmaf(acc[t][r], xs[j], bias); v = ys[j] == 0.f ? 0.f : v; }
                if constexpr (STATS) {
                    const float d = v - pvt;
                    st1 += d;
                    st2 = fmaf(d, d, st2);
                }
                if constexpr (BNB) {
                    const float xh = (yv[j] - bmu) * bis;
                    const float z = fmaf(xh, bga, bbe);
                    const float dz = v * ((z > 0.f && z < ep.bn_hi) ? 1.f : (z > 0.f ? 0.f : ep.bn_neg));
                    st1 += dz;
                    st2 = fmaf(dz, xh, st2);
                }
                if (PC_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<float*>(Cb + coff));
                else *reinterpret_cast<float*>(Cb + coff) = v;
                coff += ldc4;
            }
            coff += 4u * ldc4;
            yoff += 8u * n4;
        }
    }
    if constexpr (STATS || BNB) {
        st1 += __shfl_xor(st1, 32, 64);
        st2 += __shfl_xor(st2, 32, 64);
        if (hi == 0) {
            const int64_t rb = mw0 >> 7;                    // 128-row block index (tsii_pw_stat_rows)
            if constexpr (BNB) {
                float* sp = ep.bn_part + rb * 2 * N;
                sp[col] = st1;
                sp[N + col] = st2;
            } else {
                float* sp = ep.stats + rb * 4 * N;
                sp[col] = 128.f;
                sp[N + col] = pvt;
                sp[2 * N + col] = st1;
                sp[3 * N + col] = st2;
            }
        }
    }
}

// ABL (compile time; TSII_GEMM_PC_ABL, tools/pc_probe.py only, results are garbage): 16 no A-fragment reads, 32 no producer LDS
// stores, 64 no producer global loads, 128 no epilogue, 256 no B-fragment loads, 512 no row-scale / row-factor loads (A only),
// 1024 producer loads issued but never waited for
template <int WM, int WN, int PRODUCTS, bool BNIN, int EPI, int ABL = 0>
__global__ __launch_bounds__(768, 3) void gemm_nt_pc_kernel(const float* __restrict__ A, int64_t lda, RowScale as,
                                                            const unsigned short* __restrict__ Bp, float* __restrict__ C, int64_t ldc,
                                                            int64_t M, int N, int K, Epilogue ep, InBN ib, unsigned ntn, unsigned tiles, int opt) {
    // opt: wave priorities (A/B knob TSII_GEMM_PC_OPT; wave-uniform): bits 0-1 consumers, bits 2-3 producers
    static_assert(WM * WN == 8, "8 consumer waves");
    constexpr int P = SplitPlanes<PRODUCTS>::value;
    constexpr int BM = WM * 128, BN = WN * 32;
    constexpr int ASTAGE = P * BM * 64;                       // bytes of one LDS stage (A only)
    constexpr int R = (BM == 128) ? 5 : (BNIN ? 2 : 3);       // LDS stages
    constexpr int DA = (ABL & 16384) ? 2 : (BM == 128) ? 5 : 3;   // stages of A loads in flight per producer lane
    constexpr int NA = BM / 64;                               // A items (row, 8-k chunk) per producer thread and stage
    constexpr int LT = (ABL & 512) ? NA * 2 : NA * 4 + 2;     // counted loads per producer thread and stage
    constexpr int SLOTS = (BM == 128) ? 8 : 6;                // ring of per-tile epilogue row factors (> R: the producers run at most R tiles ahead)
    constexpr int SIDE_FLOATS = SLOTS * 2 * BM;               // [slot][X, Y][row]
    constexpr int BNV_FLOATS = BNIN ? 2048 : 0;               // input BatchNorm (scale, shift) of all K <= 1024 channels
    static_assert(DA * LT - 2 <= 63 && DA * LT >= 4, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) unsigned char smem[R * ASTAGE + (SIDE_FLOATS + BNV_FLOATS) * 4 + 64];
    float* side = reinterpret_cast<float*>(smem + R * ASTAGE);
    float* bnv = side + SIDE_FLOATS;
    unsigned* full = reinterpret_cast<unsigned*>(bnv + BNV_FLOATS);      // [8]: producer waves done with the slot, ever
    unsigned* empty = full + 8;                                          // [8]: consumer waves done with the slot, ever

    const int tid = threadIdx.x;
    // This block's tiles: tfirst + i * tstride, i < tcount.  opt bit 4 clear: a contiguous range of (row block, column block) order.
    // opt bit 4 set (round 6): dealt round robin over the blocks, with the blocks of one XCD (blockIdx % 8) taking CONSECUTIVE tiles
    // -- the column tiles of a row block, and the row blocks above / below it, then run at the same time on CUs that share an L2,
    // so the operand A (and the up-sampled addend rows, shared by two image rows) come from HBM once; in the contiguous order a CU
    // came back to them a tile (12 us) or three later and they were gone (PMC: 6.2 GB per launch against 3.4 GB algorithmic on
    // 2M x 64 -> 384).
    unsigned tfirst, tstride, tcount;
    if (opt & 16) {
        const unsigned g = gridDim.x, per = g >> 3;
        const unsigned lb = (g & 7u) == 0u ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
        tfirst = lb; tstride = g;
        tcount = (tiles - lb + g - 1) / g;                    // >= 1: the launcher never starts more blocks than tiles
    } else {
        tfirst = (unsigned)(((uint64_t)blockIdx.x * tiles) / gridDim.x);
        tstride = 1u;
        tcount = (unsigned)(((uint64_t)(blockIdx.x + 1) * tiles) / gridDim.x) - tfirst;
    }
    const unsigned t0 = tfirst;
    const int nst = (K + 31) >> 5;
    const unsigned stages = tcount * (unsigned)nst;
    constexpr bool use_cs = EPI == 2 || EPI == 3;             // dX epilogue: row factors = the two mask planes

    if (tid < 16) full[tid] = 0u;
    if constexpr (BNIN) {
        for (int i = tid; i < 2048; i += 768) {
            const int k = i & 1023;
            bnv[i] = k < K ? (i < 1024 ? ib.sc[k] : ib.sh[k]) : 0.f;
        }
    }
    __syncthreads();                                          // the only block-wide barrier

    if (tid < 512) {
        // ------------------------------------------------ consumers ------------------------------------------------
        if ((opt & 3) == 1) __builtin_amdgcn_s_setprio(1);
        else if ((opt & 3) == 2) __builtin_amdgcn_s_setprio(2);
        else if ((opt & 3) == 3) __builtin_amdgcn_s_setprio(3);
        const int cw = tid >> 6, lane = tid & 63, li = lane & 31, hi = lane >> 5;
        const int wm = cw / WN, wn = cw % WN;
        const int swz = (li >> 2) & 3;
        const int aoff = (wm * 128 + li) * 64;                          // + t * 2048 + p * BM * 64
        const int ch0 = ((0 + hi) ^ swz) << 4, ch1 = ((2 + hi) ^ swz) << 4;   // chunk of k half 0 / 1
        const char* __restrict__ Bb = reinterpret_cast<const char*>(Bp);
        const unsigned bplane = (unsigned)N * 32u;                      // bytes of one plane of one k half-step
        const unsigned bhalf = (unsigned)P * bplane;                    // bytes of one k half-step

        bf16x8 a[3][P], a3x[P], a3y[P], bx[P], by[P];     // A tiles 0-2; A tile 3 and B in two alternating sets (x: k half 0)
        f32x16 acc[4];
        auto ldf = [](const unsigned char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); };
        auto ldb = [&](const char* Bh, unsigned bl, int p) {           // Bh: wave-uniform base of the k half-step, bl: this lane's row
            return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bh + (unsigned)p * bplane + bl));
        };
        // One 16-deep k step from registers, hand-scheduled (sched_barrier pins the groups): the fragments of the NEXT step
        // (A: LDS stage Sn, chunk chn; B: half-step base Bn, lane row bln) are fetched between the MFMA groups into
        // registers that are free by then -- B and the last A tile into a second set, the other A tiles into their own
        // registers once their products are issued -- so only 3 reads trail the last 3 MFMAs.
        // (bc, a3c): B and last-A-tile fragments of the step being multiplied; (bnx, a3nx): where the next step's go.  The two
        // k halves of a stage call it with the two sets swapped, so no register copies are needed.
        auto kstep = [&](const unsigned char* Sn, int chn, const char* Bn, unsigned bln, bool pfa,
                         bf16x8 (&bc)[P], bf16x8 (&a3c)[P], bf16x8 (&bnx)[P], bf16x8 (&a3nx)[P]) {
            // pfa (wave-uniform): fetch the next A fragments too -- false in the last step of a tile (its A fragments would only
            // sit in registers across the epilogue; B, one L2 round trip away, does ride over it)
            if constexpr ((ABL & 2048) != 0) return;       // ablation: idle consumers (the producers' own pace)
            if constexpr (!(ABL & 256)) {
#pragma unroll
                for (int p = 0; p < P; ++p) bnx[p] = ldb(Bn, bln, p);
            }
            if (pfa && !(ABL & 16)) {
#pragma unroll
                for (int p = 0; p < P; ++p) a3nx[p] = ldf(Sn + aoff + 3 * 2048 + p * (BM * 64) + chn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PRODUCTS; ++q) {
                const int pa = SplitTerm<PRODUCTS>::pa(q), pb = SplitTerm<PRODUCTS>::pb(q);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][pa], bc[pb], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][pa], bc[pb], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pfa && !(ABL & 16)) {
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
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][pa], bc[pb], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3c[pa], bc[pb], acc[3], 0, 0, 0);
            }
#pragma unroll
            for (int q = QH; q < PRODUCTS; ++q)
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][SplitTerm<PRODUCTS>::pa(q)], bc[SplitTerm<PRODUCTS>::pb(q)], acc[2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (pfa && !(ABL & 16)) {
#pragma unroll
                for (int p = 0; p < P; ++p) a[2][p] = ldf(Sn + aoff + 2 * 2048 + p * (BM * 64) + chn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = QH; q < PRODUCTS; ++q)
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3c[SplitTerm<PRODUCTS>::pa(q)], bc[SplitTerm<PRODUCTS>::pb(q)], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto load_a_frags = [&](const unsigned char* Sn, int chn) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int p = 0; p < P; ++p) a[t][p] = ldf(Sn + aoff + t * 2048 + p * (BM * 64) + chn);
#pragma unroll
            for (int p = 0; p < P; ++p) a3x[p] = ldf(Sn + aoff + 3 * 2048 + p * (BM * 64) + chn);
        };
        auto lane_row = [&](unsigned tile) {                            // byte offset of this lane's B row inside a plane
            int n = (int)(tile % ntn) * BN + wn * 32 + li;
            n = n < N ? n : N - 1;
            return (unsigned)n * 32u + (unsigned)hi * 16u;
        };

        unsigned long long t_wait = 0, t_begin = 0;
        if constexpr ((ABL & 32768) != 0) t_begin = __builtin_amdgcn_s_memtime();
        unsigned slot = 0, gen1 = 4;                                    // LDS slot of the current stage; full[slot] value that means "written"
        unsigned bl = lane_row(t0);
#pragma unroll
        for (int p = 0; p < P; ++p) { bx[p] = ldb(Bb, bl, p); by[p] = bx[p]; a3y[p] = bx[p]; }      // k half-step 0 of the first tile
        // opt bits 5-6 (round 6): the SECOND consumer wave of every SIMD (waves 4-7) starts `lag` stages behind the first, so that
        // one of a SIMD's two matrix waves is in its epilogue (stores) while the other multiplies instead of both doing either
        {
            unsigned lag = ((unsigned)opt >> 5) & 3u;
            lag = lag < (unsigned)R ? lag : (unsigned)R;          // the first R stages are written without waiting for anybody
            if (lag != 0u && cw >= 4 && stages > lag) pc_wait_flag(&empty[lag - 1u], 4u);
        }
        unsigned tile = t0;
        for (unsigned ti = 0; ti < tcount; ++ti, tile += tstride) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            int colc = (int)(tile % ntn) * BN + wn * 32 + li;
            colc = colc < N ? colc : N - 1;
            const float bias = ep.bias != nullptr ? ep.bias[colc] : 0.f;                 // needed in the epilogue only: its latency is free here
            const unsigned bl_next = lane_row(ti + 1 < tcount ? tile + tstride : tile);
            pc_wait_flag(&full[slot], gen1);                            // first stage of the tile
            load_a_frags(smem + slot * ASTAGE, ch0);
            for (int ks = 0; ks < nst; ++ks) {
                const unsigned char* S0 = smem + slot * ASTAGE;
                const unsigned nslot = slot + 1 < (unsigned)R ? slot + 1 : 0u;
                const unsigned ngen1 = slot + 1 < (unsigned)R ? gen1 : gen1 + 4u;
                const char* Bh = Bb + (unsigned)(2 * ks) * bhalf;      // k half-step (ks, 0) of this tile
                const bool last = ks + 1 == nst;
                kstep(S0, ch1, Bh + bhalf, bl, true, bx, a3x, by, a3y);  // k half 0 of the stage from registers; fetch its half 1
                // every fragment read of this stage has been issued: the release below orders them before the counter
                if (lane == 0) pc_bump_flag(&empty[slot]);
                if constexpr ((ABL & 32768) != 0) {
                    const unsigned long long t0w = __builtin_amdgcn_s_memtime();
                    if (!last) pc_wait_flag(&full[nslot], ngen1);
                    t_wait += __builtin_amdgcn_s_memtime() - t0w;
                } else
                if (!last) pc_wait_flag(&full[nslot], ngen1);
                // k half 1; fetch half 0 of the next stage, or (last) only the B fragments of the next tile's first step
                kstep(smem + nslot * ASTAGE, ch0, last ? Bb : Bh + 2 * bhalf, last ? bl_next : bl, !last, by, a3y, bx, a3x);
                slot = nslot; gen1 = ngen1;
            }
            if constexpr (!(ABL & 128)) {
                const int64_t m0 = (int64_t)(tile / ntn) * BM;
                const int n0 = (int)(tile % ntn) * BN;
                const float* sx = side + (ti % SLOTS) * (2 * BM) + wm * 128;
                if (n0 + wn * 32 < N) pc_epilogue<EPI>(acc, C, ldc, N, ep, m0 + wm * 128, n0 + wn * 32, li, hi, sx, sx + BM, bias);
            } else if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) {
                C[tid] = bias;
            }
            if constexpr ((ABL & 32768) != 0) {
                if (ti + 1 == tcount && lane == 0) {
                    const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
                    C[(blockIdx.x * 8 + cw) * 2 + 0] = (float)t_wait;
                    C[(blockIdx.x * 8 + cw) * 2 + 1] = (float)t_all;
                }
            }
            bl = bl_next;
        }
    } else {
        // ------------------------------------------------ producers ------------------------------------------------
        if (((opt >> 2) & 3) == 1) __builtin_amdgcn_s_setprio(1);
        else if (((opt >> 2) & 3) == 2) __builtin_amdgcn_s_setprio(2);
        else if (((opt >> 2) & 3) == 3) __builtin_amdgcn_s_setprio(3);
        const int ptid = tid - 512, plane_lane = ptid & 63;
        const int prow = ptid >> 2, pch = ptid & 3;                     // item i: row prow + 64 i of the tile, chunk pch
        PcCursor la, wc;                                                // A loads (DA stages ahead), LDS writes
        pc_locate<BM, BN>(la, t0, ntn); la.ks = 0; la.ord = 0;
        wc = la;

        f32x4 ra[DA][NA][2];
        float sa0[DA][NA], sa1[DA][NA], sdx[DA], sdy[DA];
        const bool has_r0 = as.r0 != nullptr, has_r1 = as.r1 != nullptr;
        const float* r0p = has_r0 ? as.r0 : A;                          // branch-free: a dummy (valid) address when absent
        const float* r1p = has_r1 ? as.r1 : r0p;
        // epilogue row factors: FWD (denom, keep) / DX (cs.r0, cs.r1); absent ones read a dummy address and become 1.0
        const float* sxp = use_cs ? (ep.cs.r0 != nullptr ? ep.cs.r0 : A) : (ep.denom != nullptr ? ep.denom : A);
        const float* syp = use_cs ? (ep.cs.r1 != nullptr ? ep.cs.r1 : A) : (ep.keep != nullptr ? ep.keep : A);
        const bool has_sx = use_cs ? ep.cs.r0 != nullptr : ep.denom != nullptr, has_sy = use_cs ? ep.cs.r1 != nullptr : ep.keep != nullptr;
        const int srow = ptid < BM ? ptid : BM - 1;

        // Addresses = wave-uniform base of the tile's rows (scalar unit) + per-lane byte offsets that never change: tiles are
        // full in M, so only the k position of the stage moves the base (a k tail re-reads the row's last 8 floats; the
        // tiled B holds zeros there).
        unsigned aoffv[NA], roffv[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            aoffv[i] = (unsigned)(((prow + 64 * i) * (int)lda + pch * 8) * 4);
            if constexpr ((ABL & 4096) != 0) aoffv[i] = (unsigned)((((ptid >> 3) + 64 * i) * (int)lda + (ptid & 7) * 4) * 4);   // timing probe: full 128-byte lines per 8 lanes (wrong data)
            roffv[i] = (unsigned)((prow + 64 * i) * 4);
        }
        const unsigned soffv = (unsigned)(srow * 4);
        // the LT counted loads of one stage, always in this order: per item 2 x 16 bytes of A + 2 row scales, then the 2 row factors
        auto load_item = [&](const PcCursor& c, int slotr, int i) {
            if constexpr ((ABL & 64) != 0) { ra[slotr][i][0] = ra[slotr][i][1] = f32x4{1.f, 2.f, 3.f, 4.f}; sa0[slotr][i] = sa1[slotr][i] = 1.f; return; }
            const float* base = (ABL & 8192) ? A + (int64_t)(t0 / ntn) * BM * lda : A + c.m0 * lda + c.ks * 32;   // (8192: timing probe, every stage re-reads the first one)
            unsigned off = aoffv[i];
            if (c.ks * 32 + 32 > K) {                                   // wave-uniform: the stage holding the k tail
                const int k = c.ks * 32 + pch * 8;
                if (k > K - 8) off -= (unsigned)((k - (K - 8)) * 4);
            }
            async_load16(ra[slotr][i][0], base, off);
            async_load16(ra[slotr][i][1], base, (ABL & 4096) ? off + 32u * (unsigned)lda * 4u : off + 16u);
            if constexpr ((ABL & 512) != 0) { sa0[slotr][i] = sa1[slotr][i] = 1.f; return; }
            async_load4(sa0[slotr][i], r0p + (has_r0 ? c.m0 : 0), has_r0 ? roffv[i] : 0u);
            async_load4(sa1[slotr][i], r1p + ((has_r0 || has_r1) ? c.m0 : 0), (has_r0 || has_r1) ? roffv[i] : 0u);
        };
        auto load_side = [&](const PcCursor& c, int slotr) {
            if constexpr ((ABL & (64 | 512)) != 0) { sdx[slotr] = sdy[slotr] = 1.f; return; }
            async_load4(sdx[slotr], sxp + (has_sx ? c.m0 : 0), has_sx ? soffv : 0u);
            async_load4(sdy[slotr], syp + (has_sy ? c.m0 : 0), has_sy ? soffv : 0u);
        };
        auto store_item = [&](const PcCursor& c, unsigned char* S, int slotr, int i) {
            if constexpr ((ABL & 512) != 0 && !(ABL & 1024)) async_wait<DA * LT - 2>(ra[slotr][i][0], ra[slotr][i][1]);
            if constexpr (!(ABL & (64 | 512 | 1024))) async_wait<DA * LT - 4>(ra[slotr][i][0], ra[slotr][i][1], sa0[slotr][i], sa1[slotr][i]);
            float v[8] = {ra[slotr][i][0][0], ra[slotr][i][0][1], ra[slotr][i][0][2], ra[slotr][i][0][3],
                          ra[slotr][i][1][0], ra[slotr][i][1][1], ra[slotr][i][1][2], ra[slotr][i][1][3]};
            int k = c.ks * 32 + pch * 8;
            k = k < K - 8 ? k : K - 8;
            if constexpr (BNIN) {
                const float4 c0 = *reinterpret_cast<const float4*>(bnv + k), c1 = *reinterpret_cast<const float4*>(bnv + k + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(bnv + 1024 + k), h1 = *reinterpret_cast<const float4*>(bnv + 1024 + k + 4);
                const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                if (ib.hi < __builtin_huge_valf()) {                     // ReLU6 (wave-uniform): the upper clamp
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bn_act_load(v[e], sc[e], sh[e], ib.neg, ib.hi);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float z = fmaf(v[e], sc[e], sh[e]); v[e] = fmaxf(z, ib.neg * z); }
                }
            }
            if (has_r0) {                                               // as.split % 8 == 0 (launcher): one factor per chunk
                const float sc8 = (k < as.split) ? sa0[slotr][i] : (has_r1 ? sa1[slotr][i] : 1.f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= sc8;
            }
            u32x4 pl[P];
            split8<P>(v, pl);
            const int off = split_off(prow + 64 * i, pch);
            if constexpr ((ABL & 32) != 0) { if (pl[0][0] + pl[P - 1][3] == 0x12345u) S[0] = 1; return; }      // ablation: no LDS stores
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(S + p * (BM * 64) + off) = pl[p];
        };
        auto store_side = [&](const PcCursor& c, int slotr) {
            if constexpr (!(ABL & (64 | 512 | 1024))) async_wait<DA * LT - 2>(sdx[slotr], sdy[slotr]);
            if (c.ks == 0 && ptid < BM) {
                float* sx = side + (c.ord % SLOTS) * (2 * BM);
                const float x = has_sx ? sdx[slotr] : 1.f;
                sx[ptid] = (!use_cs && has_sx) ? 1.0f / x : x;          // one IEEE division per row
                sx[BM + ptid] = has_sy ? sdy[slotr] : 1.f;
            }
        };

        // requests of stages 0 .. DA-1 (the same order the loop re-issues them in: the wait counts hold from the first stage on)
#pragma unroll
        for (int u = 0; u < DA; ++u) {
#pragma unroll
            for (int i = 0; i < NA; ++i) load_item(la, u, i);
            load_side(la, u);
            pc_advance<BM, BN>(la, nst, ntn, tcount, tstride);
        }
        unsigned long long tp_empty = 0, tp_store = 0, tp_begin = 0, tp_wait = 0, tp_issue = 0;
        if constexpr ((ABL & 32768) != 0) tp_begin = __builtin_amdgcn_s_memtime();
        unsigned slot = 0, gen8 = 0;                                    // LDS slot of the stage being written; empty[slot] value that frees it
        for (unsigned gb = 0; gb < stages; gb += DA) {
#pragma unroll
            for (int u = 0; u < DA; ++u) {
                if (gb + u >= stages) break;
                unsigned char* S = smem + slot * ASTAGE;
                unsigned long long tq0 = 0, tq1 = 0;
                if constexpr ((ABL & 32768) != 0) tq0 = __builtin_amdgcn_s_memtime();
                if (gen8 != 0u) pc_wait_flag_lazy(&empty[slot], gen8);       // all 8 consumer waves are done with the slot's previous stage
                if constexpr ((ABL & 32768) != 0) { tq1 = __builtin_amdgcn_s_memtime(); tp_empty += tq1 - tq0; }
                if constexpr ((ABL & 32768) != 0) {       // finer timers: [vmcnt wait] [split + LDS stores] [load issue]
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const unsigned long long ta = __builtin_amdgcn_s_memtime();
                        if constexpr (!(ABL & (64 | 1024))) async_wait<DA * LT - 2>(ra[u][i][0], ra[u][i][1]);
                        const unsigned long long tb = __builtin_amdgcn_s_memtime();
                        store_item(wc, S, u, i);
                        const unsigned long long tc = __builtin_amdgcn_s_memtime();
                        load_item(la, u, i);
                        const unsigned long long td = __builtin_amdgcn_s_memtime();
                        tp_wait += tb - ta; tp_store += tc - tb; tp_issue += td - tc;
                    }
                } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) { store_item(wc, S, u, i); load_item(la, u, i); }
                }
                store_side(wc, u);
                load_side(la, u);
                if (plane_lane == 0) pc_bump_flag(&full[slot]);         // release: this wave's LDS stores precede it
                pc_advance<BM, BN>(la, nst, ntn, tcount, tstride);
                pc_advance<BM, BN>(wc, nst, ntn, tcount, tstride);
                if (++slot == (unsigned)R) { slot = 0; gen8 += 8u; }
            }
        }
        async_wait<0>(sdx[0]);                                          // nothing in flight when the wave ends
        if constexpr ((ABL & 32768) != 0) {
            if (plane_lane == 0) {
                float* d = C + 8192 + (blockIdx.x * 4 + (ptid >> 6)) * 8;
                d[0] = (float)tp_empty; d[1] = (float)tp_store; d[2] = (float)(__builtin_amdgcn_s_memtime() - tp_begin); d[3] = (float)stages;
                d[4] = (float)tp_wait; d[5] = (float)tp_issue;
            }
        }
    }
}

// ---- what was measured instead of this design (round 3; code at commit e9fb585) --------------------------------------------
// tools/probes/valu_rates.hip, mfma_valu_mix.hip on MI355X: a wave that only issues vector-ALU instructions is starved next to
// two waves of its SIMD that always have an MFMA ready (one instruction per ~10^7 cycles at equal priority; with s_setprio 3 it
// runs and the matrix pipe drops to 41-57 % busy), v_pk_{add,mul,fma}_f32 issue ~15x slower beside a busy matrix pipe (these
// files are built without them), and a wave's OWN vector-ALU instructions between its MFMAs are nearly free (6 per MFMA: pipe
// 88 % busy).  That argued for a wave-symmetric kernel (gemm_nt_ws_kernel: 8 identical waves, staging slices after every MFMA
// pair, 3 LDS stages, one LDS-only barrier per stage, every global load counted).  Built, exact (1.2e-6 vs fp64, bitwise
// repeatable), and 3-8 % SLOWER than this kernel on 11 of 14 shapes (0.69 vs 0.63 ms on 65536 x 1024 x 1024).  Its ablations
// (profiles/r03_ws_ablations.log) show why neither form gets past ~50 % of the nominal MFMA peak: the split arithmetic is free
// (0.708 -> 0.703 ms without it), and with EVERYTHING but the MFMAs removed the loop still takes 0.446 ms -- the chip is at its
// 1400 W cap (tools/clock_watch.py, profiles/r03_clock_watch.log): 2.30 GHz in the MFMA-only loop, 1.57-1.67 GHz under the full
// kernel.  Loads, LDS traffic and the epilogue cost time by costing power.

// ---- weights: fp32 [N,K] (or its transpose) -> P bf16 planes tiled as [k half-step][plane][n][16 k], zeros past K ----
template <int P>
__global__ void split_w_tiled_kernel(const float* __restrict__ w, int cols_in, int transpose, int N, int K, int nhalf,
                                     unsigned short* __restrict__ planes) {
    const int64_t total = (int64_t)nhalf * N * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 15);
        const int n = (int)((i >> 4) % N), s = (int)((i >> 4) / N);
        const int k = s * 16 + kk;
        float x = 0.f;
        if (k < K) x = transpose ? w[(int64_t)k * cols_in + n] : w[(int64_t)n * cols_in + k];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const __bf16 h = (__bf16)x;                               // RNE
            const unsigned short u = __builtin_bit_cast(unsigned short, h);
            planes[(((int64_t)s * P + p) * N + n) * 16 + kk] = u;
            x -= __builtin_bit_cast(float, (unsigned)u << 16);
        }
    }
}

#ifdef TSII_HIP_EMU
static int g_emu_cus = 0;        // TEST-ONLY: the grid size of the persistent kernel in the emulator (0: the emulator's default "device")
extern "C" void tsii_emu_set_pc_cus(int v) { g_emu_cus = v > 0 ? v : 0; }
#endif
static int pc_cus() {            // read-only device-properties cache
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus = v;
    }
#ifdef TSII_HIP_EMU
    if (g_emu_cus > 0) return g_emu_cus;
#endif
    return cus;
}

// Dispatch constants.  They are compile-time in the stock library; an A/B build (-DTSII_GEMM_PC_ABLATIONS, tools/variants) reads
// them from the environment once at load instead (tools/pc_probe.py, tools/gemm_bench.py).
#ifdef TSII_GEMM_PC_ABLATIONS
static int g_pc = getenv("TSII_GEMM_PC") ? atoi(getenv("TSII_GEMM_PC")) : 1;              // 0 = 4-wave kernels only
static int g_pc_opt = getenv("TSII_GEMM_PC_OPT") ? atoi(getenv("TSII_GEMM_PC_OPT")) : PC_OPT_DEFAULT;   // wave priorities (kernel comment)
static int g_pc_abl = getenv("TSII_GEMM_PC_ABL") ? atoi(getenv("TSII_GEMM_PC_ABL")) : 0;   // tools/pc_probe.py ablations
static int g_pc_bnb_min_k = getenv("TSII_GEMM_PC_BNB_MIN_K") ? atoi(getenv("TSII_GEMM_PC_BNB_MIN_K")) : 32;
static int g_pc_min_n = getenv("TSII_GEMM_PC_MIN_N") ? atoi(getenv("TSII_GEMM_PC_MIN_N")) : 128;
static int g_pc_wide_k = getenv("TSII_GEMM_PC_WIDE_K") ? atoi(getenv("TSII_GEMM_PC_WIDE_K")) : PC_WIDE_K;   // reductions up to this length take 128 x 256 tiles whatever N
#elif defined(TSII_HIP_EMU)
// TEST-ONLY (emulator build, tests/emu): the kernel's `opt` word (tile dealing, consumer lag), so that the CPU suite walks the forms
// the stock library does not select
static constexpr int g_pc = 1;
static int g_pc_opt = PC_OPT_DEFAULT;
extern "C" void tsii_emu_set_pc_opt(int v) { g_pc_opt = v >= 0 ? v : PC_OPT_DEFAULT; }
static constexpr int g_pc_bnb_min_k = 32, g_pc_min_n = 128, g_pc_wide_k = PC_WIDE_K;
#else
static constexpr int g_pc = 1, g_pc_opt = PC_OPT_DEFAULT;
static constexpr int g_pc_bnb_min_k = 32;   // dX + K6c: shortest reduction the persistent kernel takes (one-stage tiles: 1.57 -> 1.45 ms on 2M x 32 -> 384 since the epilogue prefetches the BatchNorm input; 64 before)
static constexpr int g_pc_min_n = 128;      // measured: 64-column outputs stay faster on the 4-wave kernel
static constexpr int g_pc_wide_k = PC_WIDE_K;
#endif

size_t nt_pc_ws_bytes(int n, int k) { return (size_t)3 * n * ((k + 31) & ~31) * sizeof(unsigned short) + 16; }

// plain 1x1 layers in the 6-product mode whose tiles are full: M a multiple of the tile height, N of 32 (everything else
// stays on the 4-wave kernels of gemm_split.hip)
static bool pc_wide(int N) { return cdiv(N, 256) * 256 == cdiv(N, 128) * 128; }      // 256-column tiles waste no more columns than 128-column ones
bool nt_pc_ok(const float* A, int64_t lda, const RowScale& as, int64_t M, int N, int K, const Epilogue& ep, const InBN& ib) {
    if (!g_pc || gemm_products() != 6) return false;
    if (N < g_pc_min_n || N % 32 != 0 || M % (pc_wide(N) ? 128 : 256) != 0 || K % 8 != 0 || lda % 4 != 0 || !aligned16(A)) return false;
    const bool dx = ep.cs.r0 != nullptr || ep.bn_y != nullptr;
    // one row-scale factor per 8-k chunk: a split inside a chunk (with r1, or with the factor-1 tail of a premultiplied second part)
    // would scale the chunk's tail by r0 -- those layers stay on the 4-wave kernels
    if (as.r0 != nullptr && as.split % 8 != 0 && as.split < K) return false;
    if (dx && (ep.denom != nullptr || ep.keep != nullptr || ep.bias != nullptr || ep.stats != nullptr || ib.sc != nullptr)) return false;   // one epilogue mode at a time
    if (ep.bn_y != nullptr && K < g_pc_bnb_min_k) return false;                                                   // (A/B knob)
    if (ib.sc != nullptr && K > 1024) return false;                                                               // (scale, shift) live in LDS
    if (ep.up_add != nullptr && (dx || ib.sc != nullptr || ep.up_w % 4 != 0 || M >= (1ll << 31) || (M / 4) * N * 4 >= (1ll << 32))) return false;   // forward only; 32-bit addend offsets
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
    TSII_REQUIRE(ldc * 128 * 4 < (1ll << 31) && (int64_t)N * 128 * 4 < (1ll << 31), "gemm_nt_pc: row pitch too large");
    const unsigned grid = (unsigned)(tiles < pc_cus() ? tiles : pc_cus());
#define TSII_PC_LAUNCH(BNINV, EPIV, ABLV) hipLaunchKernelGGL((gemm_nt_pc_kernel<WM, WN, 6, BNINV, EPIV, ABLV>), dim3(grid), dim3(768), 0, stream, \
                                                             A, lda, as, Bp, C, ldc, M, N, K, ep, ib, ntn, (unsigned)tiles, g_pc_opt)
    if (ep.bn_y != nullptr) TSII_PC_LAUNCH(false, 3, 0);
    else if (ep.cs.r0 != nullptr) TSII_PC_LAUNCH(false, 2, 0);
    else if (ep.up_add != nullptr) { if (ep.stats != nullptr) TSII_PC_LAUNCH(false, 5, 0); else TSII_PC_LAUNCH(false, 4, 0); }
    else if (ib.sc != nullptr) { if (ep.stats != nullptr) TSII_PC_LAUNCH(true, 1, 0); else TSII_PC_LAUNCH(true, 0, 0); }
    else if (ep.stats != nullptr) TSII_PC_LAUNCH(false, 1, 0);
    else {
#ifdef TSII_GEMM_PC_ABLATIONS          // A/B builds only (tools/pc_probe.py): the stock library carries none of these instantiations
        if (WM == 1 && g_pc_abl == 16) TSII_PC_LAUNCH(false, 0, 16);
        else if (WM == 1 && g_pc_abl == 32) TSII_PC_LAUNCH(false, 0, 32);
        else if (WM == 1 && g_pc_abl == 128) TSII_PC_LAUNCH(false, 0, 128);
        else if (WM == 1 && g_pc_abl == 256) TSII_PC_LAUNCH(false, 0, 256);
        else if (WM == 1 && g_pc_abl == 272) TSII_PC_LAUNCH(false, 0, 272);
        else if (WM == 1 && g_pc_abl == 400) TSII_PC_LAUNCH(false, 0, 400);
        else if (WM == 1 && g_pc_abl == 432) TSII_PC_LAUNCH(false, 0, 432);
        else if (WM == 1 && g_pc_abl == 464) TSII_PC_LAUNCH(false, 0, 464);
        else if (WM == 1 && g_pc_abl == 496) TSII_PC_LAUNCH(false, 0, 496);
        else if (WM == 1 && g_pc_abl == 96) TSII_PC_LAUNCH(false, 0, 96);
        else if (WM == 1 && g_pc_abl == 912) TSII_PC_LAUNCH(false, 0, 912);
        else if (WM == 1 && g_pc_abl == 1424) TSII_PC_LAUNCH(false, 0, 1424);
        else if (WM == 1 && g_pc_abl == 512) TSII_PC_LAUNCH(false, 0, 512);
        else if (WM == 1 && g_pc_abl == 2448) TSII_PC_LAUNCH(false, 0, 2448);
        else if (WM == 1 && g_pc_abl == 5008) TSII_PC_LAUNCH(false, 0, 5008);
        else if (WM == 1 && g_pc_abl == 9104) TSII_PC_LAUNCH(false, 0, 9104);
        else if (WM == 1 && g_pc_abl == 33680) TSII_PC_LAUNCH(false, 0, 33680);
        else if (WM == 1 && g_pc_abl == 33232) TSII_PC_LAUNCH(false, 0, 33232);
        else if (WM == 1 && g_pc_abl == 17296) TSII_PC_LAUNCH(false, 0, 17296);
        else if (WM == 1 && g_pc_abl == 4096) TSII_PC_LAUNCH(false, 0, 4096);
        else if (WM == 1 && g_pc_abl == 2512) TSII_PC_LAUNCH(false, 0, 2512);
        else
#endif
        TSII_PC_LAUNCH(false, 0, 0);
    }
#undef TSII_PC_LAUNCH
    return check_launch("gemm_nt_pc");
}

// B = fp32 [N,K] (b_transposed: fp32 [K,N]); wsplit: nt_pc_ws_bytes(N, K) bytes
int launch_nt_pc(const float* A, int64_t lda, RowScale as, const float* B, int64_t ldb, bool b_transposed, float* C, int64_t ldc,
                 int64_t M, int N, int K, Epilogue ep, InBN ib, void* wsplit, hipStream_t stream) {
    unsigned short* planes = reinterpret_cast<unsigned short*>((reinterpret_cast<uintptr_t>(wsplit) + 15) & ~(uintptr_t)15);
    const int nhalf = ((K + 31) >> 5) * 2;
    hipLaunchKernelGGL(split_w_tiled_kernel<3>, dim3(stream_grid((int64_t)nhalf * N * 16, 256)), dim3(256), 0, stream, B, (int)ldb, b_transposed ? 1 : 0, N, K, nhalf, planes);
    int rc = check_launch("split_w_tiled");
    if (rc) return rc;
    if (pc_wide(N) || (K <= g_pc_wide_k && M % 128 == 0)) return launch_nt_pc_cfg<1, 8>(A, lda, as, planes, C, ldc, M, N, K, ep, ib, stream);
    return launch_nt_pc_cfg<2, 4>(A, lda, as, planes, C, ldc, M, N, K, ep, ib, stream);
}

}  // namespace tsii
