// K4b: stems -- odd k x k, stride 2, pad (k-1)/2 convolutions over very few input channels (the 7x7 s2 3->64 first layer
// of ImageFill, models/image_inpainting.py:23; the 3x3 / 5x5 s2 stems of the other nets) -- as a stride-1 "valid"
// convolution over the space-to-depth image:
//     y[o] = sum_ky w[ky] xm[2o - pad + ky],   index 2o + ky of the zero-padded image = 2 (o + a) + p,  ky = 2a + p
// so with x2[q][(py,px,ci)] = xm_padded[2q + p][ci] (4*Cin channels, a multiple of 4) and the ceil(k/2)^2-tap kernel
// w2[(a)][(p,ci)] = w[ci][2a+p] (0 where 2a+p = k) the stem runs on the vector-gather implicit GEMM (16-byte loads, K =
// ka*ka*4*Cin) instead of the element-wise gather that decodes every (tap, channel) and fetches scalars (1.15 ms forward /
// 1.9 ms dW for ImageFill's stem).  x*mask is multiplied in while the image is rearranged; the count division and hole
// zeroing stay in the GEMM epilogue.  The input gradient is not produced here (stems sit on the data).
#include "tsii_common.h"

namespace tsii {

// x [n,h,w,c] (* mask) -> x2 [n, (h+2pad)/2, (w+2pad)/2, 4c], channel order (py, px, ci)
__global__ void stem_s2d_kernel(const float* __restrict__ x, const float* __restrict__ mfull, RowScale rs, int n, int h, int w, int c,
                                int pad, float* __restrict__ out) {
    const int h2 = (h + 2 * pad) / 2, w2 = (w + 2 * pad) / 2, c4 = 4 * c;
    const int64_t total = (int64_t)n * h2 * w2 * c4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx % c4);
        const int64_t q = idx / c4;
        const int qx = (int)(q % w2), qy = (int)((q / w2) % h2);
        const int64_t b = q / ((int64_t)w2 * h2);
        const int ci = e % c, ph = e / c;
        const int iy = 2 * qy + (ph >> 1) - pad, ix = 2 * qx + (ph & 1) - pad;
        float v = 0.f;
        if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
            const int64_t ipix = (b * h + iy) * w + ix;
            v = x[ipix * c + ci];
            if (mfull != nullptr) v *= mfull[ipix * c + ci];
            else v *= row_scale_at(rs, ipix, ci);
        }
        out[idx] = v;
    }
}

// w [cout][cin][k][k] -> w2 [cout][4 cin][ka][ka] (reference layout of the space-to-depth conv), ka = (k+1)/2
__global__ void stem_w_fwd_kernel(const float* __restrict__ w, int cout, int cin, int k, float* __restrict__ w2) {
    const int ka = (k + 1) / 2, c4 = 4 * cin;
    const int total = cout * c4 * ka * ka;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ax = i % ka, ay = (i / ka) % ka;
        const int e = (i / (ka * ka)) % c4, co = i / (ka * ka * c4);
        const int ci = e % cin, ph = e / cin;
        const int ky = 2 * ay + (ph >> 1), kx = 2 * ax + (ph & 1);
        w2[i] = (ky < k && kx < k) ? w[((co * cin + ci) * k + ky) * k + kx] : 0.f;
    }
}

// dw2 [cout][4 cin][ka][ka] -> dw [cout][cin][k][k]
__global__ void stem_w_bwd_kernel(const float* __restrict__ dw2, int cout, int cin, int k, float* __restrict__ dw) {
    const int ka = (k + 1) / 2, c4 = 4 * cin;
    const int total = cout * cin * k * k;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int kx = i % k, ky = (i / k) % k;
        const int ci = (i / (k * k)) % cin, co = i / (k * k * cin);
        const int e = ((ky & 1) * 2 + (kx & 1)) * cin + ci;
        dw[i] = dw2[((co * c4 + e) * ka + ky / 2) * ka + kx / 2];
    }
}

}  // namespace tsii

using namespace tsii;

extern "C" int tsii_stem_s2d(const float* x, const float* mfull, const float* r0, int split, const float* r1,
                             int n, int h, int w, int c, int pad, float* out, void* stream) {
    TSII_REQUIRE(x && out, "stem_s2d: null pointer");
    TSII_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && pad >= 0 && (h + 2 * pad) % 2 == 0 && (w + 2 * pad) % 2 == 0,
                 "stem_s2d: padded size must be even (h=%d w=%d pad=%d)", h, w, pad);
    const RowScale rs = {r0, r1, split};
    const int64_t total = (int64_t)n * ((h + 2 * pad) / 2) * ((w + 2 * pad) / 2) * 4 * c;
    hipLaunchKernelGGL(stem_s2d_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, mfull, rs, n, h, w, c, pad, out);
    return check_launch("stem_s2d");
}

extern "C" int tsii_stem_w_fwd(const float* w, int cout, int cin, int k, float* w2, void* stream) {
    TSII_REQUIRE(w && w2 && cout > 0 && cin > 0 && k > 0 && (k & 1), "stem_w_fwd: bad arguments");
    const int ka = (k + 1) / 2;
    hipLaunchKernelGGL(stem_w_fwd_kernel, dim3(stream_grid((int64_t)cout * 4 * cin * ka * ka, 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, k, w2);
    return check_launch("stem_w_fwd");
}

extern "C" int tsii_stem_w_bwd(const float* dw2, int cout, int cin, int k, float* dw, void* stream) {
    TSII_REQUIRE(dw2 && dw && cout > 0 && cin > 0 && k > 0 && (k & 1), "stem_w_bwd: bad arguments");
    hipLaunchKernelGGL(stem_w_bwd_kernel, dim3(stream_grid((int64_t)cout * cin * k * k, 256)), dim3(256), 0, (hipStream_t)stream, dw2, cout, cin, k, dw);
    return check_launch("stem_w_bwd");
}
