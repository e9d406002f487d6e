// K7: nearest x2 up-sampling + channel concat with the encoder skip
// (DoubleUpSample, models/partial_convolution.py:229-231 + torch.cat, models/image_inpainting.py:82-85).
// One streaming pass writes the concatenated NHWC tensor; the mask side of the same operation
// never materialises (masks stay [N,H,W] planes, see mask.hip / tsii_plane_upsample2x).
#include "tsii_common.h"

namespace tsii {

template <int W>
__global__ void upcat_fwd_kernel(const float* __restrict__ low, const float* __restrict__ skip, int n, int h, int w,
                                 int c1, int c2, float* __restrict__ out) {
    const int C = c1 + c2, CG = C / W;
    const int h2 = 2 * h, w2 = 2 * w;
    const int64_t total = (int64_t)n * h2 * w2 * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int x = (int)(pix % w2);
        const int y = (int)((pix / w2) % h2);
        const int64_t b = pix / ((int64_t)w2 * h2);
        VecF<W> v;
        if (c < c1) v = vload<W>(low + ((b * h + (y >> 1)) * w + (x >> 1)) * c1 + c);      // read 4 times: stays cached
        else v = vload_nt<W>(skip + pix * c2 + (c - c1));
        vstore_nt<W>(out + pix * C + c, v);
    }
}

// scalar form for channel counts that are not multiples of 4 (the 32 + 3 channel input of the output layer): one
// output row per blockIdx.y, so the only division left per element is a 32-bit one by C done in floating point
// (the flat-index form spent five 64-bit divisions per element: 1.05 ms for 1.4 GB)
__global__ __launch_bounds__(256) void upcat_fwd_row_kernel(const float* __restrict__ low, const float* __restrict__ skip, int h, int w,
                                                            int c1, int c2, float* __restrict__ out) {
    const int C = c1 + c2, w2 = 2 * w;
    const int64_t row = blockIdx.y;                    // (b, y) of the output
    const int y = (int)(row % (2 * h));
    const int64_t b = row / (2 * h);
    const int len = w2 * C;
    const float inv_c = 1.0f / (float)C;
    const float* __restrict__ lrow = low + (b * h + (y >> 1)) * (int64_t)w * c1;
    const float* __restrict__ srow = skip + row * (int64_t)w2 * c2;
    float* __restrict__ orow = out + row * (int64_t)len;
#pragma unroll 4
    for (int e = blockIdx.x * 1024 + threadIdx.x; e < len && e < (int)(blockIdx.x + 1) * 1024; e += 256) {
        int x = (int)((float)e * inv_c);
        if (x * C > e) --x;
        else if ((x + 1) * C <= e) ++x;
        const int c = e - x * C;
        orow[e] = c < c1 ? lrow[(x >> 1) * c1 + c] : srow[x * c2 + (c - c1)];
    }
}

template <int W>
__global__ void upcat_bwd_low_kernel(const float* __restrict__ dout, int n, int h, int w, int c1, int c2,
                                     float* __restrict__ dlow) {
    const int C = c1 + c2, CG = c1 / W;
    const int w2 = 2 * w;
    const int64_t total = (int64_t)n * h * w * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int x = (int)(pix % w);
        const int y = (int)((pix / w) % h);
        const int64_t b = pix / ((int64_t)w * h);
        const int64_t p00 = (b * 2 * h + 2 * y) * w2 + 2 * x;
        const VecF<W> a0 = vload_nt<W>(dout + p00 * C + c);
        const VecF<W> a1 = vload_nt<W>(dout + (p00 + 1) * C + c);
        const VecF<W> a2 = vload_nt<W>(dout + (p00 + w2) * C + c);
        const VecF<W> a3 = vload_nt<W>(dout + (p00 + w2 + 1) * C + c);
        VecF<W> s;
#pragma unroll
        for (int i = 0; i < W; ++i) s.v[i] = (a0.v[i] + a1.v[i]) + (a2.v[i] + a3.v[i]);
        vstore<W>(dlow + pix * c1 + c, s);
    }
}

template <int W>
__global__ void upcat_bwd_skip_kernel(const float* __restrict__ dout, int64_t npix, int c1, int c2,
                                      float* __restrict__ dskip) {
    const int C = c1 + c2, CG = c2 / W;
    const int64_t total = npix * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        vstore_nt<W>(dskip + pix * c2 + c, vload_nt<W>(dout + pix * C + c1 + c));
    }
}

// dlow[n, y, x, :] = sum over the 2 x 2 block of dout[n, 2y + dy, 2x + dx, :] * scale[n, 2y + dy, 2x + dx]  (scale: a per-pixel
// plane or NULL): the gradient of an up-sampled ADDEND (gemm_tiles.h: Epilogue::up_add) -- the adjoint of nearest x2 applied to
// the gradient that reaches the accumulator, dy * inv.  One output vector per thread, four streaming reads.
template <int W>
__global__ void pool2x2_scaled_kernel(const float* __restrict__ dout, const float* __restrict__ scale, int n, int h, int w, int c,
                                      float* __restrict__ dlow) {
    const int CG = c / W;
    const int w2 = 2 * w;
    const int64_t total = (int64_t)n * h * w * CG;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % CG) * W;
        const int64_t pix = idx / CG;
        const int x = (int)(pix % w);
        const int y = (int)((pix / w) % h);
        const int64_t b = pix / ((int64_t)w * h);
        const int64_t p00 = (b * 2 * h + 2 * y) * w2 + 2 * x;
        const VecF<W> a0 = vload_nt<W>(dout + p00 * c + cc);
        const VecF<W> a1 = vload_nt<W>(dout + (p00 + 1) * c + cc);
        const VecF<W> a2 = vload_nt<W>(dout + (p00 + w2) * c + cc);
        const VecF<W> a3 = vload_nt<W>(dout + (p00 + w2 + 1) * c + cc);
        float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f;
        if (scale != nullptr) { s0 = scale[p00]; s1 = scale[p00 + 1]; s2 = scale[p00 + w2]; s3 = scale[p00 + w2 + 1]; }
        VecF<W> s;
#pragma unroll
        for (int i = 0; i < W; ++i) s.v[i] = (a0.v[i] * s0 + a1.v[i] * s1) + (a2.v[i] * s2 + a3.v[i] * s3);
        vstore<W>(dlow + pix * c + cc, s);
    }
}

}  // namespace tsii

using namespace tsii;

extern "C" int tsii_pool2x2_scaled(const float* dout, const float* scale, int n, int h, int w, int c, float* dlow, void* stream) {
    TSII_REQUIRE(dout && dlow, "pool2x2_scaled: null pointer");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, "pool2x2_scaled: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && aligned16(dout) && aligned16(dlow);
    const int64_t total = (int64_t)n * h * w * (vec ? c / 4 : c);
    if (vec) hipLaunchKernelGGL((pool2x2_scaled_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, scale, n, h, w, c, dlow);
    else hipLaunchKernelGGL((pool2x2_scaled_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, scale, n, h, w, c, dlow);
    return check_launch("pool2x2_scaled");
}

extern "C" int tsii_upcat_fwd(const float* low, const float* skip, int n, int h, int w, int c1, int c2, float* out,
                              void* stream) {
    TSII_REQUIRE(low && out && (skip || c2 == 0), "upcat_fwd: null pointer");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && c1 > 0 && c2 >= 0, "upcat_fwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c1 % 4 == 0) && (c2 % 4 == 0) && aligned16(low) && (c2 == 0 || aligned16(skip)) && aligned16(out);
    const int64_t total = (int64_t)n * 4 * h * w * (vec ? (c1 + c2) / 4 : (c1 + c2));
    if (vec) hipLaunchKernelGGL((upcat_fwd_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, st, low, skip, n, h, w, c1, c2, out);
    else if (c2 > 0 && (int64_t)2 * w * (c1 + c2) < (1 << 24) && (int64_t)n * 2 * h <= 65535)
        hipLaunchKernelGGL(upcat_fwd_row_kernel, dim3(cdiv(2 * w * (c1 + c2), 1024), n * 2 * h), dim3(256), 0, st, low, skip, h, w, c1, c2, out);
    else hipLaunchKernelGGL((upcat_fwd_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, st, low, skip, n, h, w, c1, c2, out);
    return check_launch("upcat_fwd");
}

extern "C" int tsii_upcat_bwd(const float* dout, int n, int h, int w, int c1, int c2, float* dlow, float* dskip,
                              void* stream) {
    TSII_REQUIRE(dout, "upcat_bwd: null pointer");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && c1 > 0 && c2 >= 0, "upcat_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c1 % 4 == 0) && (c2 % 4 == 0) && aligned16(dout) && (!dlow || aligned16(dlow)) && (!dskip || aligned16(dskip));
    int rc = 0;
    if (dlow != nullptr) {
        const int64_t total = (int64_t)n * h * w * (vec ? c1 / 4 : c1);
        if (vec) hipLaunchKernelGGL((upcat_bwd_low_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, n, h, w, c1, c2, dlow);
        else hipLaunchKernelGGL((upcat_bwd_low_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, n, h, w, c1, c2, dlow);
        rc = check_launch("upcat_bwd_low");
        if (rc) return rc;
    }
    if (dskip != nullptr && c2 > 0) {
        const int64_t npix = (int64_t)n * 4 * h * w;
        const int64_t total = npix * (vec ? c2 / 4 : c2);
        if (vec) hipLaunchKernelGGL((upcat_bwd_skip_kernel<4>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, npix, c1, c2, dskip);
        else hipLaunchKernelGGL((upcat_bwd_skip_kernel<1>), dim3(flat_grid(total, 256)), dim3(256), 0, st, dout, npix, c1, c2, dskip);
        rc = check_launch("upcat_bwd_skip");
    }
    return rc;
}
