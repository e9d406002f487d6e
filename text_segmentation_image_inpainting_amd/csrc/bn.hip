// K6: BatchNorm2d (+ activation, + residual add) forward / backward on [M, C] NHWC activations
// (nn.Sequential(BatchNorm2d, act) of models/partial_convolution.py:193-197; residual adds of
// models/MobileNetV2.py:186-187 and models/image_inpainting.py:216).
//
// HBM-bound.  Statistics: task (partial row r, 4-channel group) strides over rows r, r+R, ... with
// 16-byte loads; sums are taken about a per-channel pivot (row 0) so that E[d^2]-E[d]^2 does not
// cancel, partial rows are combined in fp64.  Apply / backward-apply are single streaming passes.
#include "tsii_common.h"

namespace tsii {

template <int W>
__global__ void bn_stats_partial_kernel(const float* __restrict__ y, int64_t M, int C, int R,
                                        float* __restrict__ part) {
    const int CG = C / W;
    const int64_t tasks = (int64_t)R * CG;
    for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < tasks; task += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(task % CG) * W;
        const int r = (int)(task / CG);
        const VecF<W> piv = vload<W>(y + c);
        float s1[W], s2[W];
#pragma unroll
        for (int i = 0; i < W; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        // four rows requested before the first is used (round 6): narrow tensors have few tasks -- 4096 row lanes x C / 4 quads: two
        // waves per CU at 32 channels -- and one load in flight per thread ran them at 1.8 TB/s; same summation order as before
        int64_t m = r;
        for (; m + 3 * (int64_t)R < M; m += 4 * (int64_t)R) {
            VecF<W> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = vload<W>(y + (m + u * (int64_t)R) * C + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float d = v[u].v[i] - piv.v[i];
                    s1[i] += d;
                    s2[i] = fmaf(d, d, s2[i]);
                }
        }
        for (; m < M; m += R) {
            const VecF<W> v = vload<W>(y + m * C + c);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float d = v.v[i] - piv.v[i];
                s1[i] += d;
                s2[i] = fmaf(d, d, s2[i]);
            }
        }
        float* p = part + (int64_t)r * 2 * C;
#pragma unroll
        for (int i = 0; i < W; ++i) { p[c + i] = s1[i]; p[C + c + i] = s2[i]; }
    }
}

// combine the R partial rows: block = 32 channels x 8 row lanes, 4 independent loads in flight per lane
template <typename T>
__device__ __forceinline__ void reduce_part_rows(const T* __restrict__ part, int R, int C, int c, int ty,
                                                 double& o1, double& o2) {
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int r = ty; r < R; r += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + 8 * u;
            if (rr < R) {
                a1[u] += (double)part[(int64_t)rr * 2 * C + c];
                a2[u] += (double)part[(int64_t)rr * 2 * C + C + c];
            }
        }
    }
    o1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    o2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
}

// Level 1 of the partial-row combine: the partial passes leave up to 4096 rows of [2][C]; folding them in C/32 blocks is
// a serial chain of ~100 dependent loads (measured 54 us per BatchNorm, 2.4 ms per step), so 64-row chunks are first
// folded grid-wide into fp64 and the final kernels see at most 64 rows.
static constexpr int BN_L1_ROWS = 64;
__global__ __launch_bounds__(256) void part2_l1_kernel(const float* __restrict__ part, int R, int C, double* __restrict__ out) {
    __shared__ double sh[2][8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int r0 = blockIdx.y * BN_L1_ROWS;
    const int r1 = r0 + BN_L1_ROWS < R ? r0 + BN_L1_ROWS : R;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = r0 + ty; r < r1; r += 8) {
            a1 += (double)part[(int64_t)r * 2 * C + c];
            a2 += (double)part[(int64_t)r * 2 * C + C + c];
        }
    }
    sh[0][ty][tx] = a1; sh[1][ty][tx] = a2;
    __syncthreads();
    if (ty == 0 && c < C) {
        a1 = 0.0; a2 = 0.0;
        for (int j = 0; j < 8; ++j) { a1 += sh[0][j][tx]; a2 += sh[1][j][tx]; }
        out[(int64_t)blockIdx.y * 2 * C + c] = a1;
        out[(int64_t)blockIdx.y * 2 * C + C + c] = a2;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ y, const T* __restrict__ part,
                                                             int R, int64_t M, int C, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum) {
    __shared__ double sh[2][8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) reduce_part_rows(part, R, C, c, ty, s1, s2);
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int j = 0; j < 8; ++j) { s1 += sh[0][j][tx]; s2 += sh[1][j][tx]; }
        const double e1 = s1 / (double)M;
        double v = s2 / (double)M - e1 * e1;
        if (v < 0.0) v = 0.0;
        const double mu = (double)y[c] + e1;
        mean[c] = (float)mu;
        var[c] = (float)v;
        if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        if (running_var != nullptr) {
            const double unb = M > 1 ? v * (double)M / (double)(M - 1) : v;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

// K6b: statistics from the partials the producing conv kernel left behind: per block of rows (128-row GEMM block /
// 8x16 stencil tile; up to 16 K blocks) and channel (count n, pivot p, s1 = sum(y-p), s2 = sum((y-p)^2)) with p a
// value of the block itself, so s2 - s1^2/n does not cancel even for near-constant channels.  Level 1 turns each
// block into (n*mu_b, M2_b, n*mu_b^2) and folds 128-block chunks in fp64; the final kernel combines them
// (var = (sum M2_b + sum n mu_b^2)/M - mean^2 in fp64), updates the running statistics and emits the
// (scale, shift) pair the consumer applies on load.
// (n mu_b and M2_b + n mu_b^2 need no division: n mu_b = n p + s1 and M2_b + n mu_b^2 = s2 + 2 p s1 + n p^2 -- the s1^2 / n terms
// cancel.)  Round 4: every thread requests all its 16 rows before the first use (the loop of dependent 4-load groups made a
// 256-row finalize take 16 us, as long as the 16384-row one), and up to 1024 rows are combined by ONE kernel.
__device__ __forceinline__ void bn_part_fold(const float n, const float p, const float s1, const float s2, double& a, double& bq) {
    const double dn = (double)n, dp = (double)p, d1 = (double)s1;
    a += dn * dp + d1;
    bq += (double)s2 + dp * (2.0 * d1 + dn * dp);
}

__global__ __launch_bounds__(256) void bn_parts_l1_kernel(const float* __restrict__ part, int64_t R, int C, int chunk_rows,
                                                          double* __restrict__ out) {
    __shared__ double sh[2][8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * chunk_rows;
    const int64_t r1 = r0 + chunk_rows < R ? r0 + chunk_rows : R;
    double a = 0.0, bq = 0.0;
    if (c < C) {
        for (int64_t rb = r0 + ty; rb < r1; rb += 8 * 16) {
            float v[16][4];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t r = rb + 8 * u;
                const float* pr = part + (r < r1 ? r : r0) * 4 * C + c;
                v[u][0] = r < r1 ? pr[0] : 0.f; v[u][1] = pr[C]; v[u][2] = pr[2 * (int64_t)C]; v[u][3] = pr[3 * (int64_t)C];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (v[u][0] > 0.f) bn_part_fold(v[u][0], v[u][1], v[u][2], v[u][3], a, bq);
        }
    }
    sh[0][ty][tx] = a; sh[1][ty][tx] = bq;
    __syncthreads();
    if (ty == 0 && c < C) {
        a = 0.0; bq = 0.0;
        for (int j = 0; j < 8; ++j) { a += sh[0][j][tx]; bq += sh[1][j][tx]; }
        double* o = out + (int64_t)blockIdx.y * 3 * C + c;
        o[0] = a; o[C] = bq; o[2 * (int64_t)C] = 0.0;
    }
}

__device__ __forceinline__ void bn_finish(double a, double bq, int64_t M, int c, float* __restrict__ mean, float* __restrict__ var,
                                          float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                          float* __restrict__ scale, float* __restrict__ shift) {
    const double mu = a / (double)M;
    double v = bq / (double)M - mu * mu;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)mu;
    var[c] = (float)v;
    if (running_mean != nullptr) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    if (running_var != nullptr) {
        const double unb = M > 1 ? v * (double)M / (double)(M - 1) : v;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    if (scale != nullptr) {
        const float sc = (1.0f / sqrtf((float)v + eps)) * gamma[c];
        scale[c] = sc;
        shift[c] = beta[c] - (float)mu * sc;
    }
}

// R <= 1024 partial rows: 32 channels x 32 row lanes per block, <= 32 rows per thread in batches of 8 rows in flight
__global__ __launch_bounds__(1024) void bn_parts_small_kernel(const float* __restrict__ part, int R, int64_t M, int C,
                                                              float* __restrict__ mean, float* __restrict__ var,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float momentum, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sh[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double a = 0.0, bq = 0.0;
    if (c < C) {
        for (int rb = ty; rb < R; rb += 32 * 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 32 * u;
                const float* pr = part + (int64_t)(r < R ? r : 0) * 4 * C + c;
                v[u][0] = r < R ? pr[0] : 0.f; v[u][1] = pr[C]; v[u][2] = pr[2 * (int64_t)C]; v[u][3] = pr[3 * (int64_t)C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u][0] > 0.f) bn_part_fold(v[u][0], v[u][1], v[u][2], v[u][3], a, bq);
        }
    }
    sh[0][ty][tx] = a; sh[1][ty][tx] = bq;
    __syncthreads();
    if (ty == 0 && c < C) {
        a = 0.0; bq = 0.0;
        for (int j = 0; j < 32; ++j) { a += sh[0][j][tx]; bq += sh[1][j][tx]; }
        bn_finish(a, bq, M, c, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift);
    }
}

__global__ __launch_bounds__(256) void bn_parts_final_kernel(const double* __restrict__ l1, int chunks, int64_t M, int C,
                                                             float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sh[2][8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double a = 0.0, bq = 0.0;
    if (c < C)
        for (int r = ty; r < chunks; r += 8) {
            const double* p = l1 + (int64_t)r * 3 * C + c;
            a += p[0]; bq += p[C];
        }
    sh[0][ty][tx] = a; sh[1][ty][tx] = bq;
    __syncthreads();
    if (ty == 0 && c < C) {
        a = 0.0; bq = 0.0;
        for (int j = 0; j < 8; ++j) { a += sh[0][j][tx]; bq += sh[1][j][tx]; }
        bn_finish(a, bq, M, c, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift);
    }
}

__global__ void bn_scale_shift_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = (1.0f / sqrtf(var[c] + eps)) * gamma[c];
        scale[c] = sc;
        shift[c] = beta[c] - mean[c] * sc;
    }
}

#ifndef BN_APPLY_RPT
#define BN_APPLY_RPT 4   // rows per thread of the forward / backward apply on the large tensors (A/B: tools/variants)
#endif
// flat grid over (row blocks of RPT rows) x (channel vectors): a thread fetches its four constant vectors (and takes its rsqrt) once
// per RPT rows -- per row they were four times the streamed bytes of a plain apply (round 5, like bn_bwd_apply_kernel)
template <int W, int RPT>
__global__ void bn_act_fwd_kernel(const float* __restrict__ y, int64_t M, int C, const float* __restrict__ mean,
                                  const float* __restrict__ var, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float eps, int act, float slope,
                                  const float* __restrict__ residual, float* __restrict__ out) {
    const unsigned CG = (unsigned)(C / W);
    const int64_t total = ((M + RPT - 1) / RPT) * CG;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % CG) * W;
    const int64_t r0 = (gt / CG) * RPT;
    float mu[W], sc[W], be[W];
    {   // per-channel constants as vector loads
        const VecF<W> m4 = vload<W>(mean + c), v4 = vload<W>(var + c), g4 = vload<W>(gamma + c), b4 = vload<W>(beta + c);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            mu[i] = m4.v[i];
            sc[i] = (1.0f / sqrtf(v4.v[i] + eps)) * g4.v[i];
            be[i] = b4.v[i];
        }
    }
    VecF<W> v[RPT], res[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int64_t row = r0 + r < M ? r0 + r : M - 1;
        v[r] = vload_nt<W>(y + row * C + c);
        if (residual != nullptr) res[r] = vload_nt<W>(residual + row * C + c);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (r0 + r >= M) break;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            float z = (v[r].v[i] - mu[i]) * sc[i] + be[i];
            z = apply_act(z, act, slope);
            if (residual != nullptr) z += res[r].v[i];
            v[r].v[i] = z;
        }
        vstore_nt<W>(out + (r0 + r) * C + c, v[r]);
    }
}

// backward pass 1: per-channel s1 = sum dz, s2 = sum dz * xhat, dz = dout * act'(z)
template <int W>
__global__ void bn_bwd_partial_kernel(const float* __restrict__ dout, const float* __restrict__ y, int64_t M, int C,
                                      const float* __restrict__ mean, const float* __restrict__ var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      int act, float slope, int R, float* __restrict__ part) {
    const int CG = C / W;
    const int64_t tasks = (int64_t)R * CG;
    for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < tasks; task += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(task % CG) * W;
        const int r = (int)(task / CG);
        float mu[W], istd[W], ga[W], be[W], s1[W], s2[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            mu[i] = mean[c + i]; istd[i] = 1.0f / sqrtf(var[c + i] + eps);
            ga[i] = gamma[c + i]; be[i] = beta[c + i]; s1[i] = 0.f; s2[i] = 0.f;
        }
        // four rows (8 loads) requested before the first is used, same summation order (see bn_stats_partial_kernel)
        int64_t m = r;
        for (; m + 3 * (int64_t)R < M; m += 4 * (int64_t)R) {
            VecF<W> yv[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { yv[u] = vload<W>(y + (m + u * (int64_t)R) * C + c); dv[u] = vload<W>(dout + (m + u * (int64_t)R) * C + c); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float xh = (yv[u].v[i] - mu[i]) * istd[i];
                    const float z = xh * ga[i] + be[i];
                    const float dz = dv[u].v[i] * act_grad(z, act, slope);
                    s1[i] += dz;
                    s2[i] = fmaf(dz, xh, s2[i]);
                }
        }
        for (; m < M; m += R) {
            const VecF<W> yv = vload<W>(y + m * C + c);
            const VecF<W> dv = vload<W>(dout + m * C + c);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float xh = (yv.v[i] - mu[i]) * istd[i];
                const float z = xh * ga[i] + be[i];
                const float dz = dv.v[i] * act_grad(z, act, slope);
                s1[i] += dz;
                s2[i] = fmaf(dz, xh, s2[i]);
            }
        }
        float* p = part + (int64_t)r * 2 * C;
#pragma unroll
        for (int i = 0; i < W; ++i) { p[c + i] = s1[i]; p[C + c + i] = s2[i]; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const T* __restrict__ part, int R, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           // the backward apply's table of per-channel constants (coef[6][C]; see bn_bwd_apply_kernel)
                                                           int64_t M, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int training,
                                                           float* __restrict__ coef) {
    __shared__ double sh[2][8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) reduce_part_rows(part, R, C, c, ty, s1, s2);
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int j = 0; j < 8; ++j) { s1 += sh[0][j][tx]; s2 += sh[1][j][tx]; }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        const float invM = 1.0f / (float)M;
        coef[c] = mean[c];
        coef[C + c] = 1.0f / sqrtf(var[c] + eps);
        coef[2 * C + c] = gamma[c];
        coef[3 * C + c] = beta[c];
        coef[4 * C + c] = training ? (float)s1 * invM : 0.f;
        coef[5 * C + c] = training ? (float)s2 * invM : 0.f;
    }
}

// 65 .. 2048 partial rows in ONE launch (round 6): 32 channels x 32 row lanes per block, <= 64 rows per lane in batches of 8 in flight,
// fp64 sums -- instead of the level-1 fold + final pair (two launches of 5-6 us behind every BatchNorm backward whose K6c partial rows
// come from a GEMM epilogue or a strip kernel; ~35 per ImageFill step).  Same outputs as bn_bwd_final_kernel.
static constexpr int BN_BWD_SMALL_ROWS = 2048;
__global__ __launch_bounds__(1024) void bn_bwd_small_kernel(const float* __restrict__ part, int R, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int64_t M, const float* __restrict__ mean,
                                                            const float* __restrict__ var, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int training, float* __restrict__ coef) {
    __shared__ double sh[2][32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int rb = ty; rb < R; rb += 32 * 8) {
            float v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + 32 * u;
                const float* pr = part + (int64_t)(r < R ? r : 0) * 2 * C + c;
                v[u][0] = pr[0]; v[u][1] = pr[C];
                if (r >= R) { v[u][0] = 0.f; v[u][1] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += (double)v[u][0]; s2 += (double)v[u][1]; }
        }
    }
    sh[0][ty][tx] = s1; sh[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int j = 0; j < 32; ++j) { s1 += sh[0][j][tx]; s2 += sh[1][j][tx]; }
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        const float invM = 1.0f / (float)M;
        coef[c] = mean[c];
        coef[C + c] = 1.0f / sqrtf(var[c] + eps);
        coef[2 * C + c] = gamma[c];
        coef[3 * C + c] = beta[c];
        coef[4 * C + c] = training ? (float)s1 * invM : 0.f;
        coef[5 * C + c] = training ? (float)s2 * invM : 0.f;
    }
}

// backward pass 2: dy = gamma*istd*(dz - s1/M - xhat*s2/M)   (training)  |  gamma*istd*dz  (eval)
// A thread owns one channel vector and RPT consecutive rows (flat grid over row blocks x channel vectors): the six per-channel
// constants come from the table bn_bwd_final_kernel leaves behind (coef[j][C], j = mean, 1/std, gamma, beta, dbeta/M, dgamma/M) --
// re-deriving them per thread (24 scalar loads + 4 rsqrt) made the pass 2.3x slower than the grid-stride form it replaced -- and are
// fetched once per RPT rows: with one row per thread they were twice the streamed bytes through the vector cache (round 5; the
// bf16-storage form of this pass, where they were four times, went from 4.3 to 6 TB/s with 4 rows per thread).
template <int W, int RPT>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y, int64_t M, int C,
                                    const float* __restrict__ coef, int act, float slope, float* __restrict__ dy) {
    const unsigned CG = (unsigned)(C / W);
    const int64_t total = ((M + RPT - 1) / RPT) * CG;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % CG) * W;
    const int64_t r0 = (gt / CG) * RPT;
    const VecF<W> mu = vload<W>(coef + c), istd = vload<W>(coef + C + c), ga = vload<W>(coef + 2 * C + c), be = vload<W>(coef + 3 * C + c),
                  k1 = vload<W>(coef + 4 * C + c), k2 = vload<W>(coef + 5 * C + c);
    VecF<W> yv[RPT], dv[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {              // every load of the thread in flight before the first use
        const int64_t row = r0 + r < M ? r0 + r : M - 1;
        yv[r] = vload_nt<W>(y + row * C + c);
        dv[r] = vload_nt<W>(dout + row * C + c);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (r0 + r >= M) break;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const float xh = (yv[r].v[i] - mu.v[i]) * istd.v[i];
            const float z = xh * ga.v[i] + be.v[i];
            float dz = dv[r].v[i] * act_grad(z, act, slope);
            dz = dz - k1.v[i] - xh * k2.v[i];
            dv[r].v[i] = dz * ga.v[i] * istd.v[i];
        }
        vstore_nt<W>(dy + (r0 + r) * C + c, dv[r]);
    }
}

// The same pass over a tensor whose rows are the pixels of [.., 2h, 2w] images, ALSO leaving the 2 x 2 sums of dy * scale on the
// half-resolution grid: pooled[(n, yl, xl)] = sum_{dy,dx} dy[(n, 2yl+dy, 2xl+dx)] * scale[..] -- the gradient of an up-sampled
// addend (K7b, gemm_tiles.h: Epilogue::up_add) taken while dy is written instead of by a second pass over it
// (tsii_pool2x2_scaled: 1.0 ms per ImageFill step).  One low-resolution pixel x one channel vector per thread.
template <int W>
__global__ void bn_bwd_apply_pool_kernel(const float* __restrict__ dout, const float* __restrict__ y, int64_t Ml, int C, int hl, int wl,
                                         const float* __restrict__ coef, int act, float slope, const float* __restrict__ scale,
                                         float* __restrict__ dy, float* __restrict__ pooled) {
    const unsigned CG = (unsigned)(C / W);
    const int64_t total = Ml * CG;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= total) return;
    const int c = (int)(gt % CG) * W;
    const int64_t lp = gt / CG;                                  // (n, yl, xl)
    const int xl = (int)(lp % wl);
    const int64_t q = lp / wl;                                   // n * hl + yl
    const int64_t p00 = (2 * q) * (2 * (int64_t)wl) + 2 * xl;    // (n, 2 yl, 2 xl) on the [.., 2 hl, 2 wl] grid
    const VecF<W> mu = vload<W>(coef + c), istd = vload<W>(coef + C + c), ga = vload<W>(coef + 2 * C + c), be = vload<W>(coef + 3 * C + c),
                  k1 = vload<W>(coef + 4 * C + c), k2 = vload<W>(coef + 5 * C + c);
    const int64_t pix[4] = {p00, p00 + 1, p00 + 2 * wl, p00 + 2 * wl + 1};
    VecF<W> yv[4], dv[4];
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        yv[j] = vload_nt<W>(y + pix[j] * C + c);
        dv[j] = vload_nt<W>(dout + pix[j] * C + c);
        sc[j] = scale != nullptr ? scale[pix[j]] : 1.f;
    }
    VecF<W> acc;
#pragma unroll
    for (int i = 0; i < W; ++i) acc.v[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const float xh = (yv[j].v[i] - mu.v[i]) * istd.v[i];
            const float z = xh * ga.v[i] + be.v[i];
            float dz = dv[j].v[i] * act_grad(z, act, slope);
            dz = dz - k1.v[i] - xh * k2.v[i];
            dv[j].v[i] = dz * ga.v[i] * istd.v[i];
        }
        vstore_nt<W>(dy + pix[j] * C + c, dv[j]);
    }
    // same association as tsii_pool2x2_scaled: (a0 s0 + a1 s1) + (a2 s2 + a3 s3)
#pragma unroll
    for (int i = 0; i < W; ++i) acc.v[i] = (dv[0].v[i] * sc[0] + dv[1].v[i] * sc[1]) + (dv[2].v[i] * sc[2] + dv[3].v[i] * sc[3]);
    vstore<W>(pooled + lp * C + c, acc);
}

static int launch_bn_bwd_apply(const float* dout, const float* y, int64_t m, int c, int act, float slope, float* dy, const float* coef, hipStream_t st) {
    const bool vec = (c % 4 == 0) && aligned16(y) && aligned16(dout) && aligned16(dy) && aligned16(coef);
    const int cg = vec ? c / 4 : c;
    if (vec && m * cg >= (1ll << 21) && BN_APPLY_RPT > 1)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4, BN_APPLY_RPT>), dim3(flat_grid(cdiv64(m, BN_APPLY_RPT) * cg, 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    else if (vec) hipLaunchKernelGGL((bn_bwd_apply_kernel<4, 1>), dim3(flat_grid(m * cg, 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<1, 1>), dim3(flat_grid(m * cg, 256)), dim3(256), 0, st, dout, y, m, c, coef, act, slope, dy);
    return check_launch("bn_bwd_apply");
}

template <int W>
__global__ void act_fwd_kernel(const float* __restrict__ x, int64_t n4, int act, float slope, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        VecF<W> v = vload_nt<W>(x + i * W);
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] = apply_act(v.v[e], act, slope);
        vstore_nt<W>(out + i * W, v);
    }
}
template <int W>
__global__ void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, int64_t n4, int act,
                               float slope, float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const VecF<W> xv = vload_nt<W>(x + i * W);
        VecF<W> dv = vload_nt<W>(dout + i * W);
#pragma unroll
        for (int e = 0; e < W; ++e) dv.v[e] *= act_grad(xv.v[e], act, slope);
        vstore_nt<W>(dx + i * W, dv);
    }
}

static inline size_t bn_coef_bytes(int c) { return (size_t)6 * c * sizeof(float) + 32; }
static inline int bn_rows(int64_t m, int c) { return partial_rows(m, (c % 4 == 0) ? c / 4 : c); }

}  // namespace tsii

using namespace tsii;

extern "C" size_t tsii_bn_ws_bytes(int64_t m, int c) {
    if (m <= 0 || c <= 0) return 0;
    const size_t rows = (size_t)bn_rows(m, c);
    // partial rows + level-1 sums + the backward apply's table of 6 per-channel constants (at the end)
    return rows * 2 * c * sizeof(float) + (size_t)cdiv64((int64_t)rows, BN_L1_ROWS) * 2 * c * sizeof(double) + 16 + bn_coef_bytes(c);
}

// the constants' table sits at the end of the workspace (16-byte aligned)
static inline float* bn_coef_buffer(void* ws, size_t ws_bytes, int c) {
    uintptr_t p = (uintptr_t)ws + ws_bytes - bn_coef_bytes(c);
    return (float*)((p + 15) & ~(uintptr_t)15);
}

// doubles live after the float partial rows (8-byte aligned)
static inline double* bn_l1_buffer(void* ws, int R, int c) {
    uintptr_t p = (uintptr_t)((float*)ws + (size_t)R * 2 * c);
    return (double*)((p + 7) & ~(uintptr_t)7);
}

extern "C" int tsii_bn_stats(const float* y, int64_t m, int c, float* mean, float* var, float* running_mean,
                             float* running_var, float momentum, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(y && mean && var && ws, "bn_stats: null pointer");
    TSII_REQUIRE(m > 0 && c > 0, "bn_stats: bad shape");
    TSII_REQUIRE(ws_bytes >= tsii_bn_ws_bytes(m, c), "bn_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int R = bn_rows(m, c);
    const bool vec = (c % 4 == 0) && aligned16(y);
    const int64_t tasks = (int64_t)R * (vec ? c / 4 : c);
    float* part = (float*)ws;
    if (vec) hipLaunchKernelGGL((bn_stats_partial_kernel<4>), dim3(stream_grid(tasks, 256)), dim3(256), 0, st, y, m, c, R, part);
    else hipLaunchKernelGGL((bn_stats_partial_kernel<1>), dim3(stream_grid(tasks, 256)), dim3(256), 0, st, y, m, c, R, part);
    int rc = check_launch("bn_stats_partial");
    if (rc) return rc;
    if (R > BN_L1_ROWS) {
        double* l1 = bn_l1_buffer(ws, R, c);
        const int chunks = cdiv(R, BN_L1_ROWS);
        hipLaunchKernelGGL(part2_l1_kernel, dim3(cdiv(c, 32), chunks), dim3(256), 0, st, part, R, c, l1);
        rc = check_launch("bn_part_l1");
        if (rc) return rc;
        hipLaunchKernelGGL((bn_stats_final_kernel<double>), dim3(cdiv(c, 32)), dim3(256), 0, st, y, (const double*)l1, chunks, m, c,
                           mean, var, running_mean, running_var, momentum);
    } else {
        hipLaunchKernelGGL((bn_stats_final_kernel<float>), dim3(cdiv(c, 32)), dim3(256), 0, st, y, (const float*)part, R, m, c, mean, var,
                           running_mean, running_var, momentum);
    }
    return check_launch("bn_stats_final");
}

static inline int bn_l1_chunks(int64_t rows) { return (int)cdiv64(rows, 128); }

extern "C" size_t tsii_bn_finalize_ws_bytes(int64_t rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    return (size_t)bn_l1_chunks(rows) * 3 * c * sizeof(double);
}

extern "C" int tsii_bn_finalize(const float* stat_part, int64_t rows, int c, int64_t m,
                                float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(stat_part && mean && var && ws, "bn_finalize: null pointer");
    TSII_REQUIRE(rows > 0 && c > 0 && m > 0, "bn_finalize: bad shape");
    TSII_REQUIRE((scale == nullptr) == (shift == nullptr), "bn_finalize: scale / shift go together");
    TSII_REQUIRE(scale == nullptr || (gamma && beta), "bn_finalize: scale / shift need gamma and beta");
    TSII_REQUIRE(ws_bytes >= tsii_bn_finalize_ws_bytes(rows, c), "bn_finalize: workspace too small");
    hipStream_t st = (hipStream_t)stream;
#ifndef TSII_BN_SMALL_ROWS
#define TSII_BN_SMALL_ROWS 1024   // A/B: 0 = always the two-level path
#endif
    if (rows <= TSII_BN_SMALL_ROWS) {
        hipLaunchKernelGGL(bn_parts_small_kernel, dim3(cdiv(c, 32)), dim3(1024), 0, st, stat_part, (int)rows, m, c, mean, var, running_mean,
                           running_var, momentum, gamma, beta, eps, scale, shift);
        return check_launch("bn_parts_small");
    }
    const int chunks = bn_l1_chunks(rows);
    hipLaunchKernelGGL(bn_parts_l1_kernel, dim3(cdiv(c, 32), chunks), dim3(256), 0, st, stat_part, rows, c, 128, (double*)ws);
    int rc = check_launch("bn_parts_l1");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_parts_final_kernel, dim3(cdiv(c, 32)), dim3(256), 0, st, (const double*)ws, chunks, m, c, mean,
                       var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift);
    return check_launch("bn_parts_final");
}

extern "C" int tsii_bn_scale_shift(const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                   int c, float* scale, float* shift, void* stream) {
    TSII_REQUIRE(mean && var && gamma && beta && scale && shift && c > 0, "bn_scale_shift: bad arguments");
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, mean, var, gamma, beta, eps,
                       c, scale, shift);
    return check_launch("bn_scale_shift");
}

extern "C" int tsii_bn_act_fwd(const float* y, int64_t m, int c, const float* mean, const float* var,
                               const float* gamma, const float* beta, float eps, int act, float slope,
                               const float* residual, float* out, void* stream) {
    TSII_REQUIRE(y && mean && var && gamma && beta && out, "bn_act_fwd: null pointer");
    TSII_REQUIRE(m > 0 && c > 0, "bn_act_fwd: bad shape");
    TSII_REQUIRE(act >= 0 && act <= 4, "bn_act_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && aligned16(y) && aligned16(out) && (residual == nullptr || aligned16(residual)) &&
                     aligned16(mean) && aligned16(var) && aligned16(gamma) && aligned16(beta);
    const int cg = vec ? c / 4 : c;
    if (vec && m * cg >= (1ll << 21) && BN_APPLY_RPT > 1)
        hipLaunchKernelGGL((bn_act_fwd_kernel<4, BN_APPLY_RPT>), dim3(flat_grid(cdiv64(m, BN_APPLY_RPT) * cg, 256)), dim3(256), 0, st, y, m, c, mean, var, gamma, beta, eps, act, slope, residual, out);
    else if (vec) hipLaunchKernelGGL((bn_act_fwd_kernel<4, 1>), dim3(flat_grid(m * cg, 256)), dim3(256), 0, st, y, m, c, mean, var, gamma, beta, eps, act, slope, residual, out);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<1, 1>), dim3(flat_grid(m * cg, 256)), dim3(256), 0, st, y, m, c, mean, var, gamma, beta, eps, act, slope, residual, out);
    return check_launch("bn_act_fwd");
}

extern "C" int tsii_bn_act_bwd(const float* dout, const float* y, int64_t m, int c, const float* mean,
                               const float* var, const float* gamma, const float* beta, float eps, int act,
                               float slope, int training, float* dy, float* dgamma, float* dbeta, void* ws,
                               size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dout && y && mean && var && gamma && beta && dy && dgamma && dbeta && ws, "bn_act_bwd: null pointer");
    TSII_REQUIRE(m > 0 && c > 0, "bn_act_bwd: bad shape");
    TSII_REQUIRE(ws_bytes >= tsii_bn_ws_bytes(m, c), "bn_act_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int R = bn_rows(m, c);
    const bool vec = (c % 4 == 0) && aligned16(y) && aligned16(dout) && aligned16(dy);
    const int64_t tasks = (int64_t)R * (vec ? c / 4 : c);
    float* part = (float*)ws;
    float* coef = bn_coef_buffer(ws, ws_bytes, c);
    if (vec) hipLaunchKernelGGL((bn_bwd_partial_kernel<4>), dim3(stream_grid(tasks, 256)), dim3(256), 0, st, dout, y, m, c, mean, var, gamma, beta, eps, act, slope, R, part);
    else hipLaunchKernelGGL((bn_bwd_partial_kernel<1>), dim3(stream_grid(tasks, 256)), dim3(256), 0, st, dout, y, m, c, mean, var, gamma, beta, eps, act, slope, R, part);
    int rc = check_launch("bn_bwd_partial");
    if (rc) return rc;
    if (R > BN_L1_ROWS && R <= BN_BWD_SMALL_ROWS) {
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(cdiv(c, 32)), dim3(1024), 0, st, (const float*)part, R, c, dgamma, dbeta, m, mean, var, gamma, beta, eps, training, coef);
        rc = check_launch("bn_bwd_small");
        if (rc) return rc;
        return launch_bn_bwd_apply(dout, y, m, c, act, slope, dy, coef, st);
    }
    if (R > BN_L1_ROWS) {
        double* l1 = bn_l1_buffer(ws, R, c);
        const int chunks = cdiv(R, BN_L1_ROWS);
        hipLaunchKernelGGL(part2_l1_kernel, dim3(cdiv(c, 32), chunks), dim3(256), 0, st, part, R, c, l1);
        rc = check_launch("bn_part_l1");
        if (rc) return rc;
        hipLaunchKernelGGL((bn_bwd_final_kernel<double>), dim3(cdiv(c, 32)), dim3(256), 0, st, (const double*)l1, chunks, c, dgamma, dbeta,
                           m, mean, var, gamma, beta, eps, training, coef);
    } else {
        hipLaunchKernelGGL((bn_bwd_final_kernel<float>), dim3(cdiv(c, 32)), dim3(256), 0, st, (const float*)part, R, c, dgamma, dbeta,
                           m, mean, var, gamma, beta, eps, training, coef);
    }
    rc = check_launch("bn_bwd_final");
    if (rc) return rc;
    return launch_bn_bwd_apply(dout, y, m, c, act, slope, dy, coef, st);
}

// reductions of tsii_bn_act_bwd_pre[_pool]: partial rows -> dgamma / dbeta and the apply pass's table of per-channel constants
static int bn_bwd_pre_reduce(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int training,
                             const float* bwd_part, int64_t rows, int64_t m, int c, float* dgamma, float* dbeta, void* ws, float* coef, hipStream_t st) {
    const int R = (int)rows;
    int rc;
    if (R > BN_L1_ROWS && R <= BN_BWD_SMALL_ROWS) {
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(cdiv(c, 32)), dim3(1024), 0, st, bwd_part, R, c, dgamma, dbeta, m, mean, var, gamma, beta, eps, training, coef);
        return check_launch("bn_bwd_small");
    }
    if (R > BN_L1_ROWS) {
        double* l1 = (double*)(((uintptr_t)ws + 7) & ~(uintptr_t)7);
        const int chunks = cdiv(R, BN_L1_ROWS);
        hipLaunchKernelGGL(part2_l1_kernel, dim3(cdiv(c, 32), chunks), dim3(256), 0, st, bwd_part, R, c, l1);
        rc = check_launch("bn_part_l1");
        if (rc) return rc;
        hipLaunchKernelGGL((bn_bwd_final_kernel<double>), dim3(cdiv(c, 32)), dim3(256), 0, st, (const double*)l1, chunks, c, dgamma, dbeta,
                           m, mean, var, gamma, beta, eps, training, coef);
    } else {
        hipLaunchKernelGGL((bn_bwd_final_kernel<float>), dim3(cdiv(c, 32)), dim3(256), 0, st, bwd_part, R, c, dgamma, dbeta,
                           m, mean, var, gamma, beta, eps, training, coef);
    }
    return check_launch("bn_bwd_final");
}

// the same reduction for other translation units (bf16_elem.hip): partial rows [rows][2][c] -> dgamma / dbeta / coef[6][c];
// l1_ws: bn_bwd_reduce_ws_bytes(rows, c) bytes of scratch
namespace tsii {
size_t bn_bwd_reduce_ws_bytes(int64_t rows, int c) { return (size_t)cdiv64(rows, BN_L1_ROWS) * 2 * c * sizeof(double) + 16; }
int launch_bn_bwd_reduce(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int training,
                         const float* bwd_part, int64_t rows, int64_t m, int c, float* dgamma, float* dbeta, void* l1_ws, float* coef, hipStream_t st) {
    return bn_bwd_pre_reduce(mean, var, gamma, beta, eps, training, bwd_part, rows, m, c, dgamma, dbeta, l1_ws, coef, st);
}
}  // namespace tsii

extern "C" int tsii_bn_act_bwd_pre(const float* dout, const float* y, int64_t m, int c, const float* mean,
                                   const float* var, const float* gamma, const float* beta, float eps, int act,
                                   float slope, int training, const float* bwd_part, int64_t rows, float* dy,
                                   float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dout && y && mean && var && gamma && beta && dy && dgamma && dbeta && bwd_part && ws, "bn_act_bwd_pre: null pointer");
    TSII_REQUIRE(m > 0 && c > 0 && rows > 0 && rows < (1ll << 31), "bn_act_bwd_pre: bad shape");
    TSII_REQUIRE(ws_bytes >= (size_t)cdiv64(rows, BN_L1_ROWS) * 2 * c * sizeof(double) + 16 + bn_coef_bytes(c),
                 "bn_act_bwd_pre: workspace too small (tsii_bn_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    float* coef = bn_coef_buffer(ws, ws_bytes, c);
    int rc = bn_bwd_pre_reduce(mean, var, gamma, beta, eps, training, bwd_part, rows, m, c, dgamma, dbeta, ws, coef, st);
    if (rc) return rc;
    return launch_bn_bwd_apply(dout, y, m, c, act, slope, dy, coef, st);
}

// K6e: the two halves of tsii_bn_act_bwd_pre as entry points of their own.  tsii_bn_bwd_reduce: the K6c partial rows -> dgamma, dbeta
// and the apply pass's table coef[6][c] = (mean, 1/std, gamma, beta, dbeta/m, dgamma/m) (zeros in the last two in eval mode) -- the
// consumer that applies the BatchNorm backward while it loads (tsii_dw_bwd_dxdw_bn2) takes the table; tsii_bn_bwd_apply: the
// stand-alone apply pass over that table (what a consumer without such a form falls back to).
extern "C" size_t tsii_bn_bwd_reduce_ws_bytes(int64_t rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    return bn_bwd_reduce_ws_bytes(rows, c);
}

extern "C" int tsii_bn_bwd_reduce(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int training,
                                  const float* bwd_part, int64_t rows, int64_t m, int c, float* dgamma, float* dbeta, float* coef,
                                  void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(mean && var && gamma && beta && bwd_part && dgamma && dbeta && coef && ws, "bn_bwd_reduce: null pointer");
    TSII_REQUIRE(m > 0 && c > 0 && rows > 0 && rows < (1ll << 31), "bn_bwd_reduce: bad shape");
    TSII_REQUIRE(ws_bytes >= bn_bwd_reduce_ws_bytes(rows, c), "bn_bwd_reduce: workspace too small (tsii_bn_bwd_reduce_ws_bytes)");
    return bn_bwd_pre_reduce(mean, var, gamma, beta, eps, training, bwd_part, rows, m, c, dgamma, dbeta, ws, coef, (hipStream_t)stream);
}

extern "C" int tsii_bn_bwd_apply(const float* dout, const float* y, int64_t m, int c, const float* coef, int act, float slope, float* dy,
                                 void* stream) {
    TSII_REQUIRE(dout && y && coef && dy, "bn_bwd_apply: null pointer");
    TSII_REQUIRE(m > 0 && c > 0, "bn_bwd_apply: bad shape");
    return launch_bn_bwd_apply(dout, y, m, c, act, slope, dy, coef, (hipStream_t)stream);
}

extern "C" int tsii_bn_act_bwd_pre_pool(const float* dout, const float* y, int64_t m, int c, const float* mean,
                                        const float* var, const float* gamma, const float* beta, float eps, int act,
                                        float slope, int training, const float* bwd_part, int64_t rows,
                                        int up_h, int up_w, const float* pool_scale, float* dy, float* pooled,
                                        float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(dout && y && mean && var && gamma && beta && dy && pooled && dgamma && dbeta && bwd_part && ws, "bn_act_bwd_pre_pool: null pointer");
    TSII_REQUIRE(m > 0 && c > 0 && rows > 0 && rows < (1ll << 31), "bn_act_bwd_pre_pool: bad shape");
    TSII_REQUIRE(up_h > 0 && up_w > 0 && up_h % 2 == 0 && up_w % 2 == 0 && m % ((int64_t)up_h * up_w) == 0,
                 "bn_act_bwd_pre_pool: rows must be the pixels of whole images with even height and width (got %d x %d, m=%lld)", up_h, up_w, (long long)m);
    TSII_REQUIRE(ws_bytes >= (size_t)cdiv64(rows, BN_L1_ROWS) * 2 * c * sizeof(double) + 16 + bn_coef_bytes(c),
                 "bn_act_bwd_pre_pool: workspace too small (tsii_bn_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    float* coef = bn_coef_buffer(ws, ws_bytes, c);
    int rc = bn_bwd_pre_reduce(mean, var, gamma, beta, eps, training, bwd_part, rows, m, c, dgamma, dbeta, ws, coef, st);
    if (rc) return rc;
    const bool vec = (c % 4 == 0) && aligned16(y) && aligned16(dout) && aligned16(dy) && aligned16(pooled) && aligned16(coef);
    const int64_t ml = m / 4;
    const int64_t total = ml * (vec ? c / 4 : c);
    const unsigned grid = flat_grid(total, 256);
    if (vec) hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<4>), dim3(grid), dim3(256), 0, st, dout, y, ml, c, up_h / 2, up_w / 2, coef, act, slope, pool_scale, dy, pooled);
    else hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<1>), dim3(grid), dim3(256), 0, st, dout, y, ml, c, up_h / 2, up_w / 2, coef, act, slope, pool_scale, dy, pooled);
    return check_launch("bn_bwd_apply_pool");
}

extern "C" int tsii_act_fwd(const float* x, int64_t numel, int act, float slope, float* out, void* stream) {
    TSII_REQUIRE(x && out && numel > 0, "act_fwd: bad arguments");
    TSII_REQUIRE(act >= 0 && act <= 4, "act_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    if (numel % 4 == 0 && aligned16(x) && aligned16(out))
        hipLaunchKernelGGL((act_fwd_kernel<4>), dim3(flat_grid(numel / 4, 256)), dim3(256), 0, st, x, numel / 4, act, slope, out);
    else
        hipLaunchKernelGGL((act_fwd_kernel<1>), dim3(flat_grid(numel, 256)), dim3(256), 0, st, x, numel, act, slope, out);
    return check_launch("act_fwd");
}

extern "C" int tsii_act_bwd(const float* dout, const float* x, int64_t numel, int act, float slope, float* dx,
                            void* stream) {
    TSII_REQUIRE(dout && x && dx && numel > 0, "act_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (numel % 4 == 0 && aligned16(x) && aligned16(dout) && aligned16(dx))
        hipLaunchKernelGGL((act_bwd_kernel<4>), dim3(flat_grid(numel / 4, 256)), dim3(256), 0, st, dout, x, numel / 4, act, slope, dx);
    else
        hipLaunchKernelGGL((act_bwd_kernel<1>), dim3(flat_grid(numel, 256)), dim3(256), 0, st, dout, x, numel, act, slope, dx);
    return check_launch("act_bwd");
}
