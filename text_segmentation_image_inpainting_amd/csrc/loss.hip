// Mean-L1 loss (nn.L1Loss as used by InpaintingLoss, loss.py:190, and the throughput benchmark
// loss of SURVEY.md 8(d) cfg 2) and the fused SGD-Nesterov update the reference was trained
// with (checkpoints/ReadME.md:4).  Streaming, HBM-bound; the reduction uses wavefront shuffles.
#include "tsii_common.h"

namespace tsii {

static constexpr int L1_BLOCKS = 1024;

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         int64_t numel, float* __restrict__ part) {
    __shared__ float wsum[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        s += fabsf(a[i] - b[i]);
    s = wave_sum(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void l1_final_kernel(const float* __restrict__ part, int nblocks, int64_t numel, float* __restrict__ loss) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblocks; ++i) s += (double)part[i];
        loss[0] = (float)(s / (double)numel);
    }
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t numel,
                              const float* __restrict__ gscale, float* __restrict__ da) {
    const float g = gscale[0] / (float)numel;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        da[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
}

__global__ void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                    int64_t numel, float lr, float momentum, float wd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = fmaf(wd, p[i], g[i]);
        const float bi = fmaf(momentum, buf[i], gi);
        buf[i] = bi;
        p[i] -= lr * fmaf(momentum, bi, gi);
    }
}

}  // namespace tsii

using namespace tsii;

extern "C" size_t tsii_l1_ws_bytes(int64_t numel) { return numel > 0 ? L1_BLOCKS * sizeof(float) : 0; }

extern "C" int tsii_l1_mean_fwd(const float* a, const float* b, int64_t numel, float* loss, void* ws, size_t ws_bytes,
                                void* stream) {
    TSII_REQUIRE(a && b && loss && ws && numel > 0, "l1_mean_fwd: bad arguments");
    TSII_REQUIRE(ws_bytes >= tsii_l1_ws_bytes(numel), "l1_mean_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = cdiv64(numel, 256);
    if (nb > L1_BLOCKS) nb = L1_BLOCKS;
    hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, b, numel, (float*)ws);
    int rc = check_launch("l1_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, (int)nb, numel, loss);
    return check_launch("l1_final");
}

extern "C" int tsii_l1_mean_bwd(const float* a, const float* b, int64_t numel, const float* gscale, float* da,
                                void* stream) {
    TSII_REQUIRE(a && b && gscale && da && numel > 0, "l1_mean_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, a, b, numel, gscale, da);
    return check_launch("l1_bwd");
}

extern "C" int tsii_sgd_nesterov(float* p, const float* g, float* buf, int64_t numel, float lr, float momentum,
                                 float weight_decay, void* stream) {
    TSII_REQUIRE(p && g && buf && numel > 0, "sgd_nesterov: bad arguments");
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                       numel, lr, momentum, weight_decay);
    return check_launch("sgd_nesterov");
}
