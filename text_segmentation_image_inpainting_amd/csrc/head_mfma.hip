// K4d: weight gradient of the 3x3 output head over cat(nearest-x2(low), skip) on the f32 matrix cores.
//
// (models/image_inpainting.py:82-86: DoubleUpSample + torch.cat + the 35 -> 3 output PartialConv.  Its weight gradient is
// 8.4 M pixels x 945 multiply-adds for ImageFill's batch: 0.4 GB of HBM traffic, ~0.1 ms -- but through the vector ALU with its
// operands in LDS it took 1.36 ms per step (head_dw_kernel, dense.hip: one LDS read per ~6 FMAs and 34 % bank conflicts).)
//
//   dW[co][ci][ty][tx] = sum_q X[q][ci] * g[q - (ty-1, tx-1)][co],      g = dy * inv  (zero outside the image)
//
// is a [27 x pixels] x [pixels x cin] product with the PIXELS as the reduction dimension: rows m = (co, ty, tx) of the left
// operand are shifted reads of the small 3-channel g, the right operand is X itself, row-major as it lies in memory.
//   * up-sampled half: X[q][ci] = L[q >> 1][ci] (L = low * its mask plane) is constant over each 2 x 2 block, so the sum over a
//     block's four pixels moves into the left operand: S[j][m] = 2 x 2 box sum of g, precomputed per tile in LDS -- the matrix
//     product runs over LOW-resolution pixels, a quarter of the multiply-adds, and low is read exactly once;
//   * skip half (3 raw channels): the same product over the full-resolution pixels with the 3 channels in a 16-wide column block.
// v_mfma_f32_16x16x4_f32: D[16 x 16] += A[16 x 4] * B[4 x 16], lane l holds A[l % 16][l / 16] and B[l / 16][l % 16] -- one LDS
// dword (A) and one global dword (B; 16 consecutive channels of 4 pixels per load) per lane and instruction, exact fp32 products
// and accumulation.  Per block partial sums -> launch_reduce_rows, as for every other weight gradient here.
#include "tsii_common.h"

namespace tsii {

// ablation builds (tools/variants): 1 = no up-sampled half, 2 = no skip half, 4 = products replaced by a register sink
#ifndef HM_ABLATE
#define HM_ABLATE 0
#endif
// a value the compiler must have materialised at this point (keeps the FMAs that produce it from being sunk below later LDS
// reads, whose results then all stay live); no-op on the test emulator
#ifndef TSII_PIN_F2
#define TSII_PIN_F2(x) asm volatile("" : "+v"(x))
#endif
#ifndef HF_ABLATE
#define HF_ABLATE 0     // forward ablations: 1 = no matrix product phase, 2 = no output phase arithmetic, 4 = no skip half in it
#endif
#ifndef HM_WAVES
#define HM_WAVES 4      // resident waves per SIMD asked of the register allocator for c1 = 32 (A/B build knob)
#endif

namespace {
constexpr int HM_LH = 8, HM_LW = 32;                                  // low-resolution tile = 16 x 64 output pixels
constexpr int HM_GH = 2 * HM_LH + 2, HM_GW = 2 * HM_LW + 2;           // g with a one-pixel halo
constexpr int HM_GS = HM_GW + 1;                                      // LDS row stride (odd)
constexpr int HM_PLANE = HM_GH * HM_GS;                               // one output channel's plane
#ifdef TSII_HIP_EMU
// TEST-ONLY (emulator build, tests/emu): the block cap, so that small test tensors give a block several tiles
int g_hm_max_blocks = 1024;
#else
constexpr int g_hm_max_blocks = 1024;                                 // four resident blocks per CU
#endif

static int hm_blocks(int n, int h, int w, int* tiles_per_block) {
    const int total = n * cdiv(h / 2, HM_LH) * cdiv(w / 2, HM_LW);
    const int blocks = total < g_hm_max_blocks ? total : g_hm_max_blocks;
    *tiles_per_block = cdiv(total, blocks);
    return cdiv(total, *tiles_per_block);
}
}  // namespace

// DLOW: the same pass also writes the gradient of `low`,  dlow[j][ci] = r0l[j] * sum_m S[j][m] W[m][ci]  -- the four full-resolution
// pixels of a low pixel share every term, so it is a [low pixels x 27] x [27 x c1] product over the box sums S that are already
// in LDS for the weight gradient (head_dx_kernel, dense.hip, walks 36 accumulators per full-resolution pixel through the vector ALU:
// 0.30 ms; here 7 matrix instructions per 16 low pixels and column block).  wgt: W[co][c1 + c2][3][3].
// template <NB, R0, R1, DLOW>: 16-channel column blocks of the low tensor (c1 = 16 * NB); mask planes present; d low in the same pass
#ifndef HM_WAVES_OTHER
#define HM_WAVES_OTHER 3      // waves per SIMD of the other instantiations (c1 = 64, d low): A/B against 2 without scratch
#endif
template <int NB, bool R0, bool R1, bool DLOW>
__global__ __launch_bounds__(256, (NB == 2 && !DLOW) ? HM_WAVES : HM_WAVES_OTHER) void head_cat_dw_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                               const float* __restrict__ low, const float* __restrict__ skip,
                                                               const float* __restrict__ r0l, const float* __restrict__ r1,
                                                               const float* __restrict__ wgt, float* __restrict__ dlow,
                                                               int n, int h, int w, int c2, int cout, int tiles_per_block,
                                                               float* __restrict__ part) {
    constexpr int C1 = 16 * NB;
    static_assert(HM_LW == 32 && HM_LH == 8, "step -> pixel arithmetic below");
    __shared__ float G[3 * HM_PLANE];     // g = dy * inv, [co][row][col], halo 1
    __shared__ float T[3 * HM_PLANE];     // 2 x 2 box sums of G
    const int hl = h >> 1, wl = w >> 1;
    const int ntx = wl / HM_LW, nty = hl / HM_LH;          // whole tiles only (tsii_head_cat_low_ok)
    const int total = n * nty * ntx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nn = lane & 15, kk = lane >> 4;
    // Pixel p of step s of this wave is 4 * (wave + 4 * s) + kk = 16 * s + l4: everything about it but l4 is a compile-time
    // constant of the (unrolled) step.  LDS addresses are a lane base + an immediate, global addresses a wave-uniform base
    // (scalar arithmetic) + a lane offset that does not change from tile to tile: the products' loop has no address VALU.
    const int l4 = 4 * wave + kk;
    // left-operand rows of this lane, m = 16 * mb + nn = co * 9 + ty * 3 + tx.  Rows 27..31 of the product are never read:
    // their lanes fetch row 26's operand again rather than a zero (no select in the loop); likewise columns >= c2 of the skip block.
    int aoff_g[2], aoff_t[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = 16 * mb + nn < 27 ? 16 * mb + nn : 26;
        const int co = m / 9, t = m % 9;
        const int aoff = co * HM_PLANE + (2 - t / 3) * HM_GS + (2 - t % 3);
        aoff_g[mb] = aoff + l4;
        aoff_t[mb] = aoff + 2 * l4;
    }
    // (the copy through an opaque register right at the load keeps "uniform base + 32-bit lane offset" in that shape: the
    // compiler otherwise folds the lane offset into a 64-bit vector base and pays two vector adds per load)
    auto opq = [](unsigned o) { TSII_OPAQUE_U32(o); return o; };
    const unsigned low_lane = (unsigned)(l4 * C1 + nn) * 4u, plane_lane = (unsigned)l4 * 4u;
    const unsigned skip_lane = (unsigned)(l4 * c2 + (nn < c2 ? nn : c2 - 1)) * 4u;
    f32x4 acc_low[2][NB], acc_skip[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        acc_skip[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc_low[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // DLOW operands, fixed for the kernel, kept in LDS (as registers they would be 21 more kernel-lifetime values on top of a
    // budget that is full): row m = 4 ks + kk of the box sums (its T offset) and of W as [28][c1]
    __shared__ float WD[DLOW ? 28 * C1 : 1];
    __shared__ int DT[DLOW ? 28 : 1];
    if (DLOW) {
        for (int k = threadIdx.x; k < 28 * C1; k += 256) {
            const int m = k / C1, ci = k - m * C1;
            const int co = m / 9, t = m % 9;
            WD[k] = (m < 27 && co < cout) ? wgt[((int64_t)co * (C1 + c2) + ci) * 9 + t] : 0.f;
        }
        if (threadIdx.x < 28) {
            const int mm = threadIdx.x < 27 ? threadIdx.x : 26;
            const int co = mm / 9, t = mm % 9;
            DT[threadIdx.x] = co * HM_PLANE + (2 - t / 3) * HM_GS + (2 - t % 3);
        }
    }
    constexpr int LSTEPS = HM_LH * HM_LW / 16;        // 4-pixel steps of this wave over the tile's 256 low pixels
    constexpr int SROUNDS = 4, SSTEPS = 4 * HM_LH * HM_LW / 16 / SROUNDS;   // ... over a quarter of its 1024 full-resolution pixels
    constexpr int GIT = (HM_GH * HM_GW + 255) / 256;  // staging rounds of g
    const int t_beg = blockIdx.x * tiles_per_block;
    const int t_end = t_beg + tiles_per_block < total ? t_beg + tiles_per_block : total;
    // g of a tile (with its halo) is fetched into registers one tile ahead
    float gv[GIT][3], gs[GIT];
    auto fetch_g = [&](int tl) {
        const int bx = tl % ntx, by = (tl / ntx) % nty;
        const int64_t img = tl / (ntx * nty);
        const int y0 = 2 * HM_LH * by, x0 = 2 * HM_LW * bx;
        const float* dy_img = dy + img * h * w * cout;
        const float* inv_img = inv != nullptr ? inv + img * h * w : nullptr;
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx / HM_GW, c = idx - r * HM_GW;
            const int y = y0 - 1 + r, x = x0 - 1 + c;
            const bool ok = idx < HM_GH * HM_GW && y >= 0 && y < h && x >= 0 && x < w;
            const unsigned pix = ok ? (unsigned)(y * w + x) : 0u;
            gs[i] = ok ? 1.f : 0.f;
            if (inv != nullptr) gs[i] = ok ? inv_img[pix] : 0.f;
#pragma unroll
            for (int e = 0; e < 3; ++e) gv[i][e] = e < cout ? dy_img[pix * cout + e] : 0.f;
        }
    };
    if (t_beg < t_end) fetch_g(t_beg);
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int bx = tl % ntx, by = (tl / ntx) % nty;
        const int64_t img = tl / (ntx * nty);
        const int ly0 = by * HM_LH, lx0 = bx * HM_LW;
        const int y0 = 2 * ly0, x0 = 2 * lx0;
        // Every global load is issued well before its first consumer (a wave has nothing else to hide the latency with but the
        // other blocks of its CU): the up-sampled half's right operand for the whole tile first ...
        const char* low_t = reinterpret_cast<const char*>(low + ((img * hl + ly0) * wl + lx0) * C1);
        const char* r0_t = reinterpret_cast<const char*>(R0 ? r0l + (img * hl + ly0) * wl + lx0 : nullptr);
        const char* skip_t = reinterpret_cast<const char*>(skip + ((img * h + y0) * w + x0) * c2);
        const char* r1_t = reinterpret_cast<const char*>(R1 ? r1 + (img * h + y0) * w + x0 : nullptr);
        float bl[LSTEPS][NB], ml[R0 ? LSTEPS : 1];
#pragma unroll
        for (int u = 0; u < ((HM_ABLATE & 1) ? 0 : LSTEPS); ++u) {
            const int64_t step = (int64_t)(u >> 1) * wl + 16 * (u & 1);        // wave-uniform: low pixel (u / 2, 16 * (u % 2)) of the tile
            if (R0) ml[u] = *reinterpret_cast<const float*>(r0_t + step * 4 + opq(plane_lane));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bl[u][nb] = *reinterpret_cast<const float*>(low_t + (step * C1 + 16 * nb) * 4 + opq(low_lane));
        }
        // ... then g = dy * inv with its halo (in registers since the previous tile) -> LDS, and its 2 x 2 box sums
        __syncthreads();                              // the previous tile's operands are consumed
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int r = idx / HM_GW, c = idx - r * HM_GW;
            if (idx < HM_GH * HM_GW) {
#pragma unroll
                for (int e = 0; e < 3; ++e) G[e * HM_PLANE + r * HM_GS + c] = gv[i][e] * gs[i];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < (HM_GH - 1) * (HM_GW - 1); idx += 256) {
            const int r = idx / (HM_GW - 1), c = idx - r * (HM_GW - 1);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float* gp = G + e * HM_PLANE + r * HM_GS + c;
                T[e * HM_PLANE + r * HM_GS + c] = (gp[0] + gp[1]) + (gp[HM_GS] + gp[HM_GS + 1]);
            }
        }
        if (tl + 1 < t_end) fetch_g(tl + 1);
        // the skip half's right operand, a quarter of the tile per round, two rounds in registers: one consumed, one in flight
        float bs[2][SSTEPS], ms[2][R1 ? SSTEPS : 1];
        auto fetch_skip = [&](int rd) {
            if (HM_ABLATE & 2) return;
#pragma unroll
            for (int u = 0; u < SSTEPS; ++u) {
                const int sidx = rd * SSTEPS + u;                                 // pixel (sidx / 4, 16 * (sidx % 4) + l4) of the tile
                const int64_t step = (int64_t)(sidx / 4) * w + 16 * (sidx % 4);
                if (R1) ms[rd & 1][u] = *reinterpret_cast<const float*>(r1_t + step * 4 + opq(plane_lane));
                bs[rd & 1][u] = *reinterpret_cast<const float*>(skip_t + step * c2 * 4 + opq(skip_lane));
            }
        };
        auto skip_products = [&](int rd) {
            if (HM_ABLATE & 2) return;
#pragma unroll
            for (int u = 0; u < SSTEPS; ++u) {
                const int sidx = rd * SSTEPS + u;
                const int goff = (sidx / 4) * HM_GS + 16 * (sidx % 4);          // + l4: in aoff_g
                const float b = R1 ? bs[rd & 1][u] * ms[rd & 1][u] : bs[rd & 1][u];
                float a[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) a[mb] = G[aoff_g[mb] + goff];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    if (HM_ABLATE & 4) asm volatile("" :: "v"(a[mb]), "v"(b));
                    else acc_skip[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], b, acc_skip[mb], 0, 0, 0);
                }
                if (u % 4 == 3) __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads of later steps from piling up in registers
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        fetch_skip(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // ---- up-sampled half: 256 low pixels, 4 per instruction; wave w takes steps w, w + 4, ...
#pragma unroll
        for (int u = 0; u < ((HM_ABLATE & 1) ? 0 : LSTEPS); ++u) {
            const int toff = 2 * (u >> 1) * HM_GS + 32 * (u & 1);                                        // + 2 * l4: in aoff_t
            float a[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                a[mb] = T[aoff_t[mb] + toff];
                if (R0) a[mb] *= ml[u];                   // the lane's pixel of the left operand is its pixel of the right one
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (HM_ABLATE & 4) asm volatile("" :: "v"(a[mb]), "v"(bl[u][nb]));
                    else acc_low[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb], bl[u][nb], acc_low[mb][nb], 0, 0, 0);
                }
            if (u % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- d low: 16 low pixels per column block, wave w takes groups w, w + 4, w + 8, w + 12 (pixel (g / 2, 16 (g % 2) + lane % 16));
        //      contraction index k = 4 ks + kk <-> row m = k of the box sums and of W
        if (DLOW) {
            float* const dl_t = dlow + ((img * hl + ly0) * wl + lx0) * C1;
            int doff[7];
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) doff[ks] = DT[4 * ks + kk] + 2 * nn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = wave + 4 * r;                               // wave-uniform
                const int gorg = 2 * (g >> 1) * HM_GS + 32 * (g & 1);     // T offset of the group's first pixel
                f32x4 accd[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) accd[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const float a = T[doff[ks] + gorg];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        accd[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, WD[(4 * ks + kk) * C1 + 16 * nb + nn], accd[nb], 0, 0, 0);
                }
                // lane holds pixels 4 kk + i of the group, channel 16 nb + nn
                const int64_t prow = (int64_t)(g >> 1) * wl + 16 * (g & 1) + 4 * kk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float mk = R0 ? r0l[(img * hl + ly0) * wl + lx0 + prow + i] : 1.f;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) dl_t[(prow + i) * C1 + 16 * nb + nn] = accd[nb][i] * mk;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- skip half: 1024 full-resolution pixels, channels in the first c2 columns of one 16-wide block
#pragma unroll
        for (int rd = 0; rd < SROUNDS; ++rd) {
            if (rd + 1 < SROUNDS) fetch_skip(rd + 1);       // flies during this round's products
            __builtin_amdgcn_sched_barrier(0);
            skip_products(rd);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the four waves' sums -> one partial per block (fixed order: deterministic)
    constexpr int RS = C1 + 16 + 1;
    static_assert(32 * RS <= 3 * HM_PLANE, "block reduction must fit the g planes");
    float* red = G;
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 16 * mb + 4 * kk + i;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float* q = red + row * RS + 16 * nb + nn;
                        *q = (wv == 0 ? 0.f : *q) + acc_low[mb][nb][i];
                    }
                    float* q = red + row * RS + C1 + nn;
                    *q = (wv == 0 ? 0.f : *q) + acc_skip[mb][i];
                }
        }
    }
    __syncthreads();
    const int cin = C1 + c2;
    float* pz = part + (int64_t)blockIdx.x * cout * cin * 9;
    for (int k = threadIdx.x; k < 9 * cout * cin; k += 256) {
        const int c = k % cin, m = k / cin;        // m = co * 9 + t
        pz[((int64_t)(m / 9) * cin + c) * 9 + m % 9] = red[m * RS + c];
    }
}


// ---- forward -------------------------------------------------------------------------------------------------------------------
//   y[p][co] = sum_t ( sum_ci L[(p + t - 1) >> 1][ci] W[co][ci][t]  +  sum_cs XS[p + t - 1][cs] W[co][c1 + cs][t] )
// The inner sum of the up-sampled half depends on the LOW pixel only: Z[j][(t, co)] = sum_ci L[j][ci] W[co][ci][t] is a
// [27 x c1] x [c1 x low pixels] product on the matrix cores (a quarter of the multiply-adds of the full-resolution form, low read
// once), left in LDS for the tile with its one-pixel halo; every output pixel then adds 9 of its entries per output channel, and the
// 3-channel skip half (81 multiply-adds per pixel, weights in scalar registers) from a staged patch.  (head_fwd_kernel, dense.hip,
// runs all 945 multiply-adds per pixel through the vector ALU from LDS: 0.67 ms for ImageFill's batch, 0.1 ms of HBM traffic.)
constexpr int HF_ZH = HM_LH + 2, HF_ZW = HM_LW + 2;                   // low pixels of a tile with halo: 10 x 34
constexpr int HF_GROUPS = (HF_ZH * HF_ZW + 15) / 16;                  // 16-pixel column blocks of the product: 22
constexpr int HF_ROUNDS = (HF_GROUPS + 3) / 4;                        // per wave: 6
constexpr int HF_ZP = 356;                                            // Z row stride (4 * HF_ZP % 64 == 16: the 4 row groups of a store hit different banks)
constexpr int HF_XH = 2 * HM_LH + 2, HF_XW = 2 * HM_LW + 2, HF_XS = HF_XW + 1;   // skip patch 18 x 66, row stride 67
constexpr int HF_XIT = (HF_XH * HF_XW * 3 + 255) / 256;               // staging rounds of the skip patch (3 channels)
static_assert(HF_GROUPS * 16 <= HF_ZP, "Z row holds every column block");

__device__ __attribute__((aligned(16))) float g_zero_page[64] = {};   // what a load outside the image reads (zero padding without a select)
__device__ float g_one = 1.f;                                          // what an absent denom / keep plane reads (stride 0)

template <int NB, bool R0, bool R1>
__global__ __launch_bounds__(256, 2) void head_cat_fwd_mfma_kernel(const float* __restrict__ low, const float* __restrict__ skip,
                                                                   const float* __restrict__ r0l, const float* __restrict__ r1,
                                                                   const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                   const float* __restrict__ denom, const float* __restrict__ keep,
                                                                   int n, int h, int w, int cout, int tiles_per_block, float* __restrict__ y) {
    constexpr int C1 = 16 * NB, C2 = 3;
    const float* zeros = g_zero_page;
    constexpr int GB = NB == 2 ? HF_ROUNDS : HF_ROUNDS / 2;            // column blocks whose loads are in flight together
    static_assert(HF_ROUNDS % GB == 0, "whole batches");
    __shared__ float Z[32 * HF_ZP];                                     // rows 27..31 are written (zeros) and never read
    __shared__ float XS[3 * HF_XH * HF_XS];
    __shared__ __attribute__((aligned(16))) float WS[9 * C2 * 4];
    const int hl = h >> 1, wl = w >> 1;
    const int ntx = wl / HM_LW, nty = hl / HM_LH;
    const int total = n * nty * ntx;
    const int cin = C1 + C2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nn = lane & 15, kk = lane >> 4;
    // The contraction index of one matrix instruction is spread over the four 16-lane groups: instruction (j, e) lets group kk
    // carry channel 16 j + 4 kk + e -- so a lane's right operands for e = 0..3 are ONE 16-byte load of its pixel (channels
    // 16 j + 4 kk ..), and the left operand (fixed for the kernel) is W as [m = t * 3 + co][that channel], rows >= 27 and output
    // channels >= cout zero.
    float wl_[2][NB][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = 16 * mb + nn;
        const int t = m / 3, co = m - 3 * t;
        const bool ok = m < 27 && co < cout;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) wl_[mb][j][e] = ok ? wgt[((int64_t)co * cin + 16 * j + 4 * kk + e) * 9 + t] : 0.f;
    }
    // the skip half's 81 weights as [tap][cs][co (padded to 4)] in LDS: one broadcast 16-byte read per (tap, cs) in the output phase
    // (as 81 scalar registers they do not fit beside the rest and spill through v_readlane)
    for (int k = threadIdx.x; k < 9 * C2 * 4; k += 256) {
        const int co = k & 3, cs = (k >> 2) % C2, t = k / (4 * C2);
        WS[k] = co < cout ? wgt[((int64_t)co * cin + C1 + cs) * 9 + t] : 0.f;
    }
    const float* kbase = keep != nullptr ? keep : &g_one;
    const float* dbase = denom != nullptr ? denom : &g_one;
    const int64_t kstride = keep != nullptr ? 1 : 0, dstride = denom != nullptr ? 1 : 0;
    float bv[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bv[co] = (bias != nullptr && co < cout) ? bias[co] : 0.f;
    const int t_beg = blockIdx.x * tiles_per_block;
    const int t_end = t_beg + tiles_per_block < total ? t_beg + tiles_per_block : total;
    for (int tl = t_beg; tl < t_end; ++tl) {
        const int bx = tl % ntx, by = (tl / ntx) % nty;
        const int64_t img = tl / (ntx * nty);
        const int ly0 = by * HM_LH, lx0 = bx * HM_LW;
        const int y0 = 2 * ly0, x0 = 2 * lx0;
        const float* low_img = low + img * hl * wl * C1;
        const float* r0_img = R0 ? r0l + img * hl * wl : nullptr;
        const float* skip_img = skip + img * h * w * C2;
        const float* r1_img = R1 ? r1 + img * h * w : nullptr;
        // (the thread index goes through an opaque copy per tile: everything derived from it is tile-invariant, and hoisted out of
        // the tile loop it is ~100 registers of staging indices that spill)
        unsigned tid = threadIdx.x;
        TSII_OPAQUE_U32(tid);
        // the skip patch (with its halo), one float per thread and round: rows of 66 x 3 contiguous floats
        float xv[HF_XIT], xm[R1 ? HF_XIT : 1];
#pragma unroll
        for (int i = 0; i < HF_XIT; ++i) {
            const int idx = (int)tid + 256 * i;
            const int r = idx / (HF_XW * 3), j = idx - r * (HF_XW * 3);
            const int c = j / 3;
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            const bool ok = idx < HF_XH * HF_XW * 3 && yy >= 0 && yy < h && xx >= 0 && xx < w;
            const unsigned pix = ok ? (unsigned)(yy * w + xx) : 0u;
            const float* src = ok ? skip_img + pix * 3 + (j - 3 * c) : zeros;      // zero padding: a load of a zero
            xv[i] = *src;
            if (R1) xm[i] = r1_img[pix];
        }
        f32x4 bq[GB][NB];
        float mq[R0 ? GB : 1];
        auto fetch_low = [&](int batch) {
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                const int pg = 16 * (wave + 4 * (batch * GB + q)) + nn;
                const int ry = pg / HF_ZW, rx = pg - ry * HF_ZW;
                const int ly = ly0 - 1 + ry, lx = lx0 - 1 + rx;
                const bool ok = pg < HF_ZH * HF_ZW && ly >= 0 && ly < hl && lx >= 0 && lx < wl;
                const unsigned lp = ok ? (unsigned)(ly * wl + lx) : 0u;
                if (R0) mq[q] = r0_img[lp];
                const float* src = ok ? low_img + lp * C1 + 4 * kk : zeros;          // pixels outside the image: zeros (64 of them at `zeros`)
#pragma unroll
                for (int j = 0; j < NB; ++j) bq[q][j] = *reinterpret_cast<const f32x4*>(src + 16 * j);
            }
        };
        if (!(HF_ABLATE & 1)) fetch_low(0);
        __syncthreads();                              // the previous tile's Z and patch are consumed
        unsigned tid2 = threadIdx.x;
        TSII_OPAQUE_U32(tid2);
#pragma unroll
        for (int i = 0; i < HF_XIT; ++i) {
            const int idx = (int)tid2 + 256 * i;
            const int r = idx / (HF_XW * 3), j = idx - r * (HF_XW * 3);
            const int c = j / 3;
            if (idx < HF_XH * HF_XW * 3) XS[(j - 3 * c) * (HF_XH * HF_XS) + r * HF_XS + c] = R1 ? xv[i] * xm[i] : xv[i];
        }
        // ---- Z = W_low x L over the tile's low pixels with halo: 16 pixels per column block, wave w takes blocks w, w + 4, ...
#pragma unroll
        for (int batch = 0; batch < ((HF_ABLATE & 1) ? 0 : HF_ROUNDS / GB); ++batch) {
            if (batch > 0) fetch_low(batch);
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                const int g = wave + 4 * (batch * GB + q);
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl_[mb][j][e], bq[q][j][e], acc[mb], 0, 0, 0);
                if (g < HF_GROUPS) {                  // (wave-uniform) the lane's results all belong to its pixel: the mask goes on here
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) Z[(16 * mb + 4 * kk + i) * HF_ZP + 16 * g + nn] = R0 ? acc[mb][i] * mq[q] : acc[mb][i];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---- outputs: thread = column x of the tile, rows 4 * wave .. 4 * wave + 3 (the row arithmetic is wave-uniform)
        unsigned tid3 = threadIdx.x;
        TSII_OPAQUE_U32(tid3);
        const int x = (int)(tid3 & 63u), yq = __builtin_amdgcn_readfirstlane((int)(tid3 >> 6));
        int rxo[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) rxo[tx] = ((x + tx - 1) >> 1) + 1;
        const int64_t pix0 = (img * h + y0 + 4 * yq) * w + x0 + x;
        float kp[4], dn[4];                            // (absent planes: a 1 read with stride 0 -- no branches around the loads)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kp[i] = kbase[(pix0 + (int64_t)i * w) * kstride];
            dn[i] = dbase[(pix0 + (int64_t)i * w) * dstride];
        }
        float a[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int co = 0; co < 3; ++co) a[i][co] = 0.f;
        // skip half: per (column tap, channel) the 6 patch rows of the thread's 4 output rows are read once; the weight reads go
        // through an opaque zero so that they stay here (hoisted out of the tile loop they are 108 registers)
        unsigned zt = 0u;
        TSII_OPAQUE_U32(zt);
#pragma unroll
        for (int tx = 0; tx < ((HF_ABLATE & 6) ? 0 : 3); ++tx)
#pragma unroll
            for (int cs = 0; cs < C2; ++cs) {
                float xs[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) xs[r] = XS[cs * (HF_XH * HF_XS) + (4 * yq + r) * HF_XS + x + tx];
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(WS + ((ty * 3 + tx) * C2 + cs) * 4 + zt);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int co = 0; co < 3; ++co) a[i][co] = fmaf(xs[i + ty], wv[co], a[i][co]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int co = 0; co < 3; ++co) TSII_PIN_F2(a[i][co]);     // (else every LDS read of the phase is issued first and
                __builtin_amdgcn_sched_barrier(0);                            //  the FMAs sink below them: 270 live registers)
            }
        // up-sampled half: 9 entries of Z per output pixel and channel
#pragma unroll
        for (int i = 0; i < ((HF_ABLATE & 2) ? 0 : 4); ++i) {
            const int yy = 4 * yq + i;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const int ry = ((yy + ty - 1) >> 1) + 1;
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const float* zp = Z + ((ty * 3 + tx) * 3) * HF_ZP + ry * HF_ZW + rxo[tx];
#pragma unroll
                    for (int co = 0; co < 3; ++co) a[i][co] += zp[co * HF_ZP];
                }
            }
#pragma unroll
            for (int co = 0; co < 3; ++co) TSII_PIN_F2(a[i][co]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                if (co >= cout) break;
                float v = a[i][co] / dn[i] + bv[co];
                if (kp[i] == 0.f) v = 0.f;
                y[(pix0 + (int64_t)i * w) * cout + co] = v;
            }
    }
}

}  // namespace tsii

using namespace tsii;

extern "C" size_t tsii_dense_bwd_dw_ws_bytes(int n, int ho, int wo, int cin, int cout, int kh, int kw);

#ifdef TSII_HIP_EMU
extern "C" void tsii_emu_set_head_blocks(int v) { g_hm_max_blocks = v > 0 ? v : 1024; }
#endif

extern "C" int tsii_head_cat_low_ok(int n, int h, int wd, int c1, int c2, int cout) {
    if (n <= 0 || h <= 0 || wd <= 0 || h % (2 * HM_LH) != 0 || wd % (2 * HM_LW) != 0) return 0;     // whole 16 x 64 tiles
    if ((int64_t)h * wd * (c1 > 4 * c2 ? c1 / 4 : c2) * 4 >= (1ll << 32)) return 0;      // 32-bit byte offsets inside one image
    return (c1 == 32 || c1 == 64) && c2 >= 1 && c2 <= 16 && cout >= 1 && cout <= 3 ? 1 : 0;
}

static int head_cat_bwd_low_impl(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                 int c1, int c2, const float* r0_low, const float* r1, const float* w, int n, int h, int wd, int cout,
                                 float* dwgt, float* dbias, float* dlow, void* ws, size_t ws_bytes, void* stream);

extern "C" int tsii_head_cat_bwd_dw_low(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                        int c1, int c2, const float* r0_low, const float* r1, int n, int h, int wd, int cout,
                                        float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    return head_cat_bwd_low_impl(dy, inv, keep, low, skip, c1, c2, r0_low, r1, nullptr, n, h, wd, cout, dwgt, dbias, nullptr, ws, ws_bytes, stream);
}

extern "C" int tsii_head_cat_bwd_low(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                     int c1, int c2, const float* r0_low, const float* r1, const float* w, int n, int h, int wd, int cout,
                                     float* dwgt, float* dbias, float* dlow, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(w && dlow, "head_cat_bwd_low: null pointer");
    TSII_REQUIRE(c1 == 32, "head_cat_bwd_low: the fused d low exists for 32 low channels (tsii_head_cat_bwd_low_ok)");
    TSII_REQUIRE(aligned16(dlow), "head_cat_bwd_low: dlow must be 16-byte aligned");
    return head_cat_bwd_low_impl(dy, inv, keep, low, skip, c1, c2, r0_low, r1, w, n, h, wd, cout, dwgt, dbias, dlow, ws, ws_bytes, stream);
}

static int head_cat_bwd_low_impl(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                 int c1, int c2, const float* r0_low, const float* r1, const float* w, int n, int h, int wd, int cout,
                                 float* dwgt, float* dbias, float* dlow, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && low && skip && dwgt && ws, "head_cat_bwd_dw_low: null pointer");
    TSII_REQUIRE(tsii_head_cat_low_ok(n, h, wd, c1, c2, cout), "head_cat_bwd_dw_low: geometry has no matrix-core head (tsii_head_cat_low_ok)");
    TSII_REQUIRE(ws_bytes >= tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3), "head_cat_bwd_dw_low: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    int tpb = 0;
    const int blocks = hm_blocks(n, h, wd, &tpb);
    // the kernel writes `blocks` partial rows, the bias column sums go behind them: checked against what THIS call writes, not only
    // against tsii_dense_bwd_dw_ws_bytes()'s own row plan (the two agree today through the 1024-block caps; nothing else ties them)
    TSII_REQUIRE(((size_t)blocks * cout * (c1 + c2) * 9 + (dbias != nullptr ? colsum_ws_floats((int64_t)n * h * wd, cout) : 0)) * sizeof(float) <= ws_bytes,
                 "head_cat_bwd_dw_low: workspace too small for %d partial rows", blocks);
#define TSII_HM_LAUNCH(NB, R0, R1) do { \
    if (NB == 2 && dlow != nullptr) hipLaunchKernelGGL((head_cat_dw_mfma_kernel<2, R0, R1, true>), dim3(blocks), dim3(256), 0, st, dy, inv, low, skip, r0_low, r1, w, dlow, n, h, wd, c2, cout, tpb, part); \
    else hipLaunchKernelGGL((head_cat_dw_mfma_kernel<NB, R0, R1, false>), dim3(blocks), dim3(256), 0, st, dy, inv, low, skip, r0_low, r1, w, dlow, n, h, wd, c2, cout, tpb, part); } while (0)
#define TSII_HM_MASKS(NB)                                           \
    do {                                                            \
        if (r0_low != nullptr && r1 != nullptr) TSII_HM_LAUNCH(NB, true, true);   \
        else if (r0_low != nullptr) TSII_HM_LAUNCH(NB, true, false);              \
        else if (r1 != nullptr) TSII_HM_LAUNCH(NB, false, true);                  \
        else TSII_HM_LAUNCH(NB, false, false);                                    \
    } while (0)
    if (c1 == 32) TSII_HM_MASKS(2);
    else TSII_HM_MASKS(4);
#undef TSII_HM_MASKS
#undef TSII_HM_LAUNCH
    int rc = check_launch("head_cat_dw_mfma");
    if (rc) return rc;
    const int64_t len = (int64_t)cout * (c1 + c2) * 9;
    rc = launch_reduce_rows(part, blocks, len, dwgt, st);
    if (rc) return rc;
    if (dbias != nullptr) rc = launch_colsum_scaled(dy, keep, (int64_t)n * h * wd, cout, dbias, part + (size_t)blocks * len, st);
    return rc;
}

extern "C" int tsii_head_cat_bwd_low_ok(int n, int h, int wd, int c1, int c2, int cout) {
    return tsii_head_cat_low_ok(n, h, wd, c1, c2, cout) && c1 == 32 ? 1 : 0;
}

extern "C" int tsii_head_cat_fwd_low_ok(int n, int h, int wd, int c1, int c2, int cout) {
    return tsii_head_cat_low_ok(n, h, wd, c1, c2, cout) && c2 == 3 ? 1 : 0;
}

extern "C" int tsii_head_cat_fwd_low(const float* low, const float* skip, int c1, int c2, const float* r0_low, const float* r1,
                                     const float* w, const float* bias, const float* denom, const float* keep,
                                     int n, int h, int wd, int cout, float* y, void* stream) {
    TSII_REQUIRE(low && skip && w && y, "head_cat_fwd_low: null pointer");
    TSII_REQUIRE(aligned16(low), "head_cat_fwd_low: low must be 16-byte aligned");
    TSII_REQUIRE(tsii_head_cat_fwd_low_ok(n, h, wd, c1, c2, cout), "head_cat_fwd_low: geometry has no matrix-core head (tsii_head_cat_fwd_low_ok)");
    hipStream_t st = (hipStream_t)stream;
    int tpb = 0;
    const int blocks = hm_blocks(n, h, wd, &tpb);
#define TSII_HF_LAUNCH(NB, R0, R1) \
    hipLaunchKernelGGL((head_cat_fwd_mfma_kernel<NB, R0, R1>), dim3(blocks), dim3(256), 0, st, low, skip, r0_low, r1, w, bias, denom, keep, n, h, wd, cout, tpb, y)
#define TSII_HF_MASKS(NB)                                           \
    do {                                                            \
        if (r0_low != nullptr && r1 != nullptr) TSII_HF_LAUNCH(NB, true, true);   \
        else if (r0_low != nullptr) TSII_HF_LAUNCH(NB, true, false);              \
        else if (r1 != nullptr) TSII_HF_LAUNCH(NB, false, true);                  \
        else TSII_HF_LAUNCH(NB, false, false);                                    \
    } while (0)
    if (c1 == 32) TSII_HF_MASKS(2);
    else TSII_HF_MASKS(4);
#undef TSII_HF_MASKS
#undef TSII_HF_LAUNCH
    return check_launch("head_cat_fwd_mfma");
}
