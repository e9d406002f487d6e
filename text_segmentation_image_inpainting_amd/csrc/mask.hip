// K1: mask bookkeeping on [N,H,W] planes.  Replaces the all-ones "mask convolutions" of the
// reference (models/partial_convolution.py:41-47,57-66,74-77,129-135): the count of valid
// inputs is a box sum of the channel-summed mask, identical for every output channel
// (SURVEY.md F5).  All values are small integers in fp32 -> bit-exact in any summation order.
#include "tsii_common.h"

namespace tsii {

__global__ void mask_channel_sum_kernel(const float* __restrict__ mask, int64_t total, int h, int w, int c,
                                        int64_t sn, int64_t sh, int64_t sw, int64_t sc, float* __restrict__ plane) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const int64_t n = i / ((int64_t)w * h);
        const float* p = mask + n * sn + (int64_t)y * sh + (int64_t)x * sw;
        float s = 0.f;
        for (int ch = 0; ch < c; ++ch) s += p[(int64_t)ch * sc];
        plane[i] = s;
    }
}

__global__ void mask_update_kernel(const float* __restrict__ p0, float a0, const float* __restrict__ p1, float a1,
                                   int64_t total, int h, int w, int kh, int kw, int sh, int sw, int ph, int pw,
                                   int dh, int dw, int ho, int wo, float post_scale, int fill_holes,
                                   float* __restrict__ denom, float* __restrict__ new_mask, float* __restrict__ inv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo);
        const int oy = (int)((i / wo) % ho);
        const int64_t n = i / ((int64_t)wo * ho);
        const int64_t base = n * (int64_t)h * w;
        float cnt = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int iy = oy * sh - ph + ky * dh;
            if (iy < 0 || iy >= h) continue;  // zero padding counts as hole
            for (int kx = 0; kx < kw; ++kx) {
                const int ix = ox * sw - pw + kx * dw;
                if (ix < 0 || ix >= w) continue;
                const int64_t o = base + (int64_t)iy * w + ix;
                float s = a0 * p0[o];
                if (p1 != nullptr) s += a1 * p1[o];
                cnt += s;
            }
        }
        const bool hole = (cnt == 0.f);
        float d = cnt * post_scale, nm = 1.f, iv;
        if (fill_holes && hole) {
            d = 1.f; nm = 0.f; iv = 0.f;
        } else {
            iv = 1.f / d;
        }
        if (denom != nullptr) denom[i] = d;
        if (new_mask != nullptr) new_mask[i] = nm;
        if (inv != nullptr) inv[i] = iv;
    }
}

__global__ void plane_upsample2x_kernel(const float* __restrict__ in, int64_t total_out, int h, int w,
                                        float* __restrict__ out) {
    const int w2 = 2 * w, h2 = 2 * h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_out; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w2);
        const int y = (int)((i / w2) % h2);
        const int64_t n = i / ((int64_t)w2 * h2);
        out[i] = in[n * (int64_t)h * w + (int64_t)(y >> 1) * w + (x >> 1)];
    }
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t numel,
                           float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i];
}

}  // namespace tsii

using namespace tsii;

extern "C" int tsii_mask_channel_sum(const float* mask, int n, int h, int w, int c,
                                     int64_t sn, int64_t sh, int64_t sw, int64_t sc, float* plane, void* stream) {
    TSII_REQUIRE(mask && plane, "mask_channel_sum: null pointer");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, "mask_channel_sum: bad shape %d %d %d %d", n, h, w, c);
    const int64_t total = (int64_t)n * h * w;
    hipLaunchKernelGGL(mask_channel_sum_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       mask, total, h, w, c, sn, sh, sw, sc, plane);
    return check_launch("mask_channel_sum");
}

extern "C" int tsii_mask_update(const float* p0, float a0, const float* p1, float a1, int n, int h, int w,
                                int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int ho, int wo,
                                float post_scale, int fill_holes, float* denom, float* new_mask, float* inv,
                                void* stream) {
    TSII_REQUIRE(p0, "mask_update: null plane");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0,
                 "mask_update: bad geometry");
    TSII_REQUIRE(ho == (h + 2 * ph - dh * (kh - 1) - 1) / sh + 1 && wo == (w + 2 * pw - dw * (kw - 1) - 1) / sw + 1,
                 "mask_update: output size %dx%d inconsistent with conv geometry", ho, wo);
    const int64_t total = (int64_t)n * ho * wo;
    hipLaunchKernelGGL(mask_update_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       p0, a0, p1, a1, total, h, w, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo, post_scale, fill_holes,
                       denom, new_mask, inv);
    return check_launch("mask_update");
}

extern "C" int tsii_plane_upsample2x(const float* in, int n, int h, int w, float* out, void* stream) {
    TSII_REQUIRE(in && out && n > 0 && h > 0 && w > 0, "plane_upsample2x: bad arguments");
    const int64_t total = (int64_t)n * h * w * 4;
    hipLaunchKernelGGL(plane_upsample2x_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, total, h, w, out);
    return check_launch("plane_upsample2x");
}

extern "C" int tsii_mul_mask(const float* x, const float* mask, int64_t numel, float* out, void* stream) {
    TSII_REQUIRE(x && mask && out && numel > 0, "mul_mask: bad arguments");
    hipLaunchKernelGGL(mul_kernel, dim3(flat_grid(numel, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, mask, numel, out);
    return check_launch("mul_mask");
}
