// Error plumbing + two tiny shared kernels (partial-row reduction, 2-D transpose).
#include "tsii_common.h"

#include <stdarg.h>
#include <stdio.h>

namespace tsii {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
        return -1000 - (int)e;
    }
    return 0;
}

// ---- out[j] = sum_r ws[r][j] -------------------------------------------------
// block = 64 columns x 16 row lanes, 4 independent loads in flight per lane: 1024 partial rows are 16 dependent rounds (one thread
// per column walking the rows serially measured 52 us per call; 4 row lanes 25 us on average over the step's 36 calls, 0.9 ms).
// perm_cin / perm_T != 0: the partial rows are laid out [co][t][ci] and the output [co][ci][t] (dense-conv weight gradients from the
// implicit-GEMM kernels): reads stay contiguous, the few writes scatter.
constexpr int RR_LANES = 16;
__global__ __launch_bounds__(64 * RR_LANES) void reduce_rows_kernel(const float* __restrict__ ws, int rows, int64_t len, float* __restrict__ out,
                                                                    int perm_cin, int perm_T) {
    __shared__ double sh[RR_LANES][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int64_t j0 = (int64_t)blockIdx.x * 64; j0 < len; j0 += (int64_t)gridDim.x * 64) {
        const int64_t j = j0 + tx;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (j < len) {
            for (int r = ty; r < rows; r += 4 * RR_LANES) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + RR_LANES * u;
                    if (rr < rows) a[u] += (double)ws[(int64_t)rr * len + j];
                }
            }
        }
        __syncthreads();
        sh[ty][tx] = (a[0] + a[1]) + (a[2] + a[3]);
        __syncthreads();
        if (ty == 0 && j < len) {
            double t = 0.0;
#pragma unroll
            for (int l = 0; l < RR_LANES; ++l) t += sh[l][tx];
            int64_t dst = j;
            if (perm_T != 0) {                       // j = (co * T + t) * cin + ci  ->  (co * cin + ci) * T + t
                const int ci = (int)(j % perm_cin);
                const int64_t q = j / perm_cin;
                dst = ((q / perm_T) * perm_cin + ci) * perm_T + q % perm_T;
            }
            out[dst] = (float)t;
        }
    }
}

// wide rows (the [splits][P x Q] slabs of the split-K weight-gradient GEMMs: up to 64 x 1 MB, read at 2.5 TB/s by the kernel
// above): four columns per thread, 16-byte loads
__global__ __launch_bounds__(64 * RR_LANES) void reduce_rows_vec4_kernel(const float* __restrict__ ws, int rows, int64_t len4, float* __restrict__ out) {
    __shared__ double sh[RR_LANES][64][4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(ws);
    for (int64_t j0 = (int64_t)blockIdx.x * 64; j0 < len4; j0 += (int64_t)gridDim.x * 64) {
        const int64_t j = j0 + tx;
        double a[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        if (j < len4) {
            for (int r = ty; r < rows; r += 4 * RR_LANES) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = r + RR_LANES * u;
                    v[u] = rr < rows ? w4[(int64_t)rr * len4 + j] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[u & 1][e] += (double)v[u][e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) sh[ty][tx][e] = a[0][e] + a[1][e];
        __syncthreads();
        if (ty < 4 && j < len4) {                 // lane e of the first four adds up component e
            double t = 0.0;
#pragma unroll
            for (int l = 0; l < RR_LANES; ++l) t += sh[l][tx][ty];
            out[j * 4 + ty] = (float)t;
        }
    }
}

int launch_reduce_rows(const float* ws, int rows, int64_t len, float* out, hipStream_t stream) {
    if (len >= 4096 && len % 4 == 0 && aligned16(ws) && rows >= 8) {
        hipLaunchKernelGGL(reduce_rows_vec4_kernel, dim3(stream_grid(cdiv64(len / 4, 64) * 256, 256)), dim3(64 * RR_LANES), 0, stream, ws, rows, len / 4, out);
        return check_launch("reduce_rows_vec4");
    }
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(stream_grid(cdiv64(len, 64) * 256, 256)), dim3(64 * RR_LANES), 0, stream, ws, rows, len, out, 0, 0);
    return check_launch("reduce_rows");
}
int launch_reduce_rows_conv(const float* ws, int rows, int cout, int cin, int T, float* out, hipStream_t stream) {
    const int64_t len = (int64_t)cout * cin * T;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(stream_grid(cdiv64(len, 64) * 256, 256)), dim3(64 * RR_LANES), 0, stream, ws, rows, len, out, cin, T);
    return check_launch("reduce_rows_conv");
}

// ---- 2-D transpose through a padded 32x32 LDS tile -----------------------------
__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? in[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;  // out is [cols, rows]
        if (c < cols && r < rows) out[(int64_t)c * rows + r] = tile[tx][j];
    }
}

int launch_transpose(const float* in, int rows_in, int cols_in, float* out, hipStream_t stream) {
    dim3 grid(cdiv(cols_in, 32), cdiv(rows_in, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, rows_in, cols_in, out);
    return check_launch("transpose");
}

// ---- scaled column sums (bias gradients) -----------------------------------------
// block = NB columns x L row lanes (NB = min(N,256), L = min(1024/NB, 64)); the [rows, N] slab of a block is one
// contiguous stream, four independent rows in flight per lane (256 threads with one row at a time ran the 2 M x 64 stem gradient
// at 2 TB/s: 128 dependent memory latencies per lane); lanes are combined through LDS.  part[block][n]
__global__ __launch_bounds__(1024) void colsum_partial_kernel(const float* __restrict__ a, const float* __restrict__ rowmul,
                                                             int64_t M, int N, int NB, int L, int64_t rows_per_block,
                                                             float* __restrict__ part) {
    __shared__ float sh[1024];
    const int col0 = blockIdx.y * NB;
    const int cl = threadIdx.x % NB, lane = threadIdx.x / NB;
    const int col = col0 + cl;
    const int64_t mbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t mend = (mbeg + rows_per_block < M) ? mbeg + rows_per_block : M;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < L && col < N) {
        for (int64_t m = mbeg + lane; m < mend; m += 4 * (int64_t)L) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t mm = m + (int64_t)u * L;
                if (mm < mend) {
                    float v = a[mm * N + col];
                    if (rowmul != nullptr) v *= rowmul[mm];
                    s[u] += v;
                }
            }
        }
    }
    sh[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (lane == 0 && col < N) {
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += sh[l * NB + cl];
        part[(int64_t)blockIdx.x * N + col] = t;
    }
}

static inline int64_t colsum_rpb(int64_t M) { int64_t r = cdiv64(M, 1024); return r < 64 ? 64 : r; }
size_t colsum_ws_floats(int64_t M, int N) { return (size_t)cdiv64(M, colsum_rpb(M)) * N; }

int launch_colsum_scaled(const float* a, const float* rowmul, int64_t M, int N, float* out, float* ws,
                         hipStream_t stream) {
    const int64_t rpb = colsum_rpb(M);
    const int rows = (int)cdiv64(M, rpb);
    const int NB = N < 256 ? N : 256, L = 1024 / NB < 64 ? 1024 / NB : 64;      // <= 64 lanes: lane 0 adds them up serially
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(rows, cdiv(N, NB)), dim3(1024), 0, stream, a, rowmul, M, N, NB, L, rpb, ws);
    int rc = check_launch("colsum_partial");
    if (rc) return rc;
    return launch_reduce_rows(ws, rows, N, out, stream);
}

}  // namespace tsii

extern "C" int tsii_version(void) { return TSII_ABI_VERSION; }
extern "C" const char* tsii_last_error(void) { return tsii::g_err; }
