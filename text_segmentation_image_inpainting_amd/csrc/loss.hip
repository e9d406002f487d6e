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

// ---- InpaintingLoss pieces (loss.py:195-225,303-307) -------------------------------------
__global__ void compose_kernel(const float* __restrict__ raw, const float* __restrict__ mask, const float* __restrict__ out,
                               int64_t numel, float* __restrict__ comp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i];
        comp[i] = m * raw[i] + (1.f - m) * out[i];
    }
}
__global__ void compose_bwd_kernel(const float* __restrict__ dcomp, const float* __restrict__ mask, int64_t numel,
                                   float* __restrict__ dout) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        dout[i] = dcomp[i] * (1.f - mask[i]);
}

__device__ __forceinline__ float block_sum_256(float s, float* wsum) {
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void masked_l1_partial_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                                                const float* __restrict__ mask, int64_t numel,
                                                                float wv, float wh, float* __restrict__ part) {
    __shared__ float wsum[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i], o = out[i], g = gt[i];
        s += wv * fabsf(m * o - m * g) + wh * fabsf((1.f - m) * o - (1.f - m) * g);
    }
    const float t = block_sum_256(s, wsum);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void masked_l1_bwd_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                     const float* __restrict__ mask, int64_t numel, float wv, float wh,
                                     const float* __restrict__ gscale, float* __restrict__ dout) {
    const float gs = gscale[0] / (float)numel;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i], o = out[i], g = gt[i];
        const float dv = m * o - m * g, dh = (1.f - m) * o - (1.f - m) * g;
        const float sv = dv > 0.f ? 1.f : (dv < 0.f ? -1.f : 0.f);
        const float sh = dh > 0.f ? 1.f : (dh < 0.f ? -1.f : 0.f);
        dout[i] = gs * (wv * sv * m + wh * sh * (1.f - m));
    }
}

// part[2*block + 0] = sum |x - x_right|, part[2*block + 1] = sum |x - x_down|
__global__ __launch_bounds__(256) void tv_partial_kernel(const float* __restrict__ x, int64_t numel, int h, int w, int c,
                                                         float* __restrict__ part) {
    __shared__ float wsum[4];
    float sw = 0.f, sh = 0.f;
    const int64_t wc = (int64_t)w * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / c;
        const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
        const float v = x[i];
        if (xx + 1 < w) sw += fabsf(v - x[i + c]);
        if (yy + 1 < h) sh += fabsf(v - x[i + wc]);
    }
    const float tw = block_sum_256(sw, wsum);
    __syncthreads();
    const float th = block_sum_256(sh, wsum);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = tw; part[2 * blockIdx.x + 1] = th; }
}
__global__ void tv_final_kernel(const float* __restrict__ part, int nblocks, double cnt_w, double cnt_h, float* __restrict__ loss) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < nblocks; ++i) { a += (double)part[2 * i]; b += (double)part[2 * i + 1]; }
        loss[0] = (float)(a / cnt_w + b / cnt_h);
    }
}
__device__ __forceinline__ float sgnf(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
__global__ void tv_bwd_kernel(const float* __restrict__ x, int64_t numel, int h, int w, int c, float gw, float gh,
                              const float* __restrict__ gscale, float* __restrict__ dx) {
    const float g = gscale[0];
    const int64_t wc = (int64_t)w * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / c;
        const int xx = (int)(pix % w), yy = (int)((pix / w) % h);
        const float v = x[i];
        float d = 0.f;
        if (xx + 1 < w) d += gw * sgnf(v - x[i + c]);
        if (xx > 0) d -= gw * sgnf(x[i - c] - v);
        if (yy + 1 < h) d += gh * sgnf(v - x[i + wc]);
        if (yy > 0) d -= gh * sgnf(x[i - wc] - v);
        dx[i] = g * d;
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
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, a, b, numel, gscale, da);
    return check_launch("l1_bwd");
}

extern "C" int tsii_sgd_nesterov(float* p, const float* g, float* buf, int64_t numel, float lr, float momentum,
                                 float weight_decay, void* stream) {
    TSII_REQUIRE(p && g && buf && numel > 0, "sgd_nesterov: bad arguments");
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                       numel, lr, momentum, weight_decay);
    return check_launch("sgd_nesterov");
}

extern "C" int tsii_compose_fwd(const float* raw, const float* mask, const float* out, int64_t numel, float* comp, void* stream) {
    TSII_REQUIRE(raw && mask && out && comp && numel > 0, "compose_fwd: bad arguments");
    hipLaunchKernelGGL(compose_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, raw, mask, out, numel, comp);
    return check_launch("compose_fwd");
}

extern "C" int tsii_compose_bwd(const float* dcomp, const float* mask, int64_t numel, float* dout, void* stream) {
    TSII_REQUIRE(dcomp && mask && dout && numel > 0, "compose_bwd: bad arguments");
    hipLaunchKernelGGL(compose_bwd_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, dcomp, mask, numel, dout);
    return check_launch("compose_bwd");
}

extern "C" int tsii_masked_l1_fwd(const float* out, const float* gt, const float* mask, int64_t numel, float w_valid,
                                  float w_hole, float* loss, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(out && gt && mask && loss && ws && numel > 0, "masked_l1_fwd: bad arguments");
    TSII_REQUIRE(ws_bytes >= tsii_l1_ws_bytes(numel), "masked_l1_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = cdiv64(numel, 256);
    if (nb > L1_BLOCKS) nb = L1_BLOCKS;
    hipLaunchKernelGGL(masked_l1_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, out, gt, mask, numel, w_valid, w_hole, (float*)ws);
    int rc = check_launch("masked_l1_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, (int)nb, numel, loss);
    return check_launch("masked_l1_final");
}

extern "C" int tsii_masked_l1_bwd(const float* out, const float* gt, const float* mask, int64_t numel, float w_valid,
                                  float w_hole, const float* gscale, float* dout, void* stream) {
    TSII_REQUIRE(out && gt && mask && gscale && dout && numel > 0, "masked_l1_bwd: bad arguments");
    hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, out, gt, mask,
                       numel, w_valid, w_hole, gscale, dout);
    return check_launch("masked_l1_bwd");
}

extern "C" int tsii_tv_fwd(const float* x, int n, int h, int w, int c, float* loss, void* ws, size_t ws_bytes, void* stream) {
    TSII_REQUIRE(x && loss && ws && n > 0 && h > 1 && w > 1 && c > 0, "tv_fwd: bad arguments");
    const int64_t numel = (int64_t)n * h * w * c;
    TSII_REQUIRE(ws_bytes >= 2 * tsii_l1_ws_bytes(numel), "tv_fwd: workspace too small (need 2x tsii_l1_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = cdiv64(numel, 256);
    if (nb > L1_BLOCKS) nb = L1_BLOCKS;
    hipLaunchKernelGGL(tv_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, numel, h, w, c, (float*)ws);
    int rc = check_launch("tv_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(tv_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, (int)nb, (double)n * c * h * (w - 1),
                       (double)n * c * (h - 1) * w, loss);
    return check_launch("tv_final");
}

extern "C" int tsii_tv_bwd(const float* x, int n, int h, int w, int c, const float* gscale, float* dx, void* stream) {
    TSII_REQUIRE(x && gscale && dx && n > 0 && h > 1 && w > 1 && c > 0, "tv_bwd: bad arguments");
    const int64_t numel = (int64_t)n * h * w * c;
    const float gw = (float)(1.0 / ((double)n * c * h * (w - 1))), gh = (float)(1.0 / ((double)n * c * (h - 1) * w));
    hipLaunchKernelGGL(tv_bwd_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream, x, numel, h, w, c, gw, gh,
                       gscale, dx);
    return check_launch("tv_bwd");
}
