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

template <int NB, bool R0, bool R1>   // 16-channel column blocks of the low tensor (c1 = 16 * NB); mask planes present
__global__ __launch_bounds__(256, NB == 2 ? HM_WAVES : 3) void head_cat_dw_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ inv,
                                                               const float* __restrict__ low, const float* __restrict__ skip,
                                                               const float* __restrict__ r0l, const float* __restrict__ r1,
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

extern "C" int tsii_head_cat_bwd_dw_low(const float* dy, const float* inv, const float* keep, const float* low, const float* skip,
                                        int c1, int c2, const float* r0_low, const float* r1, int n, int h, int wd, int cout,
                                        float* dwgt, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dy && low && skip && dwgt && ws, "head_cat_bwd_dw_low: null pointer");
    TSII_REQUIRE(tsii_head_cat_low_ok(n, h, wd, c1, c2, cout), "head_cat_bwd_dw_low: geometry has no matrix-core head (tsii_head_cat_low_ok)");
    TSII_REQUIRE(ws_bytes >= tsii_dense_bwd_dw_ws_bytes(n, h, wd, c1 + c2, cout, 3, 3), "head_cat_bwd_dw_low: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    int tpb = 0;
    const int blocks = hm_blocks(n, h, wd, &tpb);
#define TSII_HM_LAUNCH(NB, R0, R1) \
    hipLaunchKernelGGL((head_cat_dw_mfma_kernel<NB, R0, R1>), dim3(blocks), dim3(256), 0, st, dy, inv, low, skip, r0_low, r1, n, h, wd, c2, cout, tpb, part)
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
