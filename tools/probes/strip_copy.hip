// Hardware probe (not part of the library): what does the MARCHING-STRIP access pattern itself cost on the MI355X?
// out = f(in) over an NHWC fp32 tensor [32][256][256][384] (3.2 GB in, 3.2 GB out).  A block owns a strip of TW pixels x CB
// channels and marches down it R rows at a time, the next step's rows requested (global -> registers) before the current ones
// are stored: the depth-wise strip kernels' traffic without their LDS, halo and arithmetic.  Variants: strip shape, rows per step,
// prefetch depth, non-temporal hints, block order.   hipcc --offload-arch=gfx950 -O3 -o strip_copy strip_copy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid & 7u, local = bid >> 3;
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ORDER 0: channel block fastest (the library's order), 1: strip fastest, 2: no XCD remap (channel block fastest)
template <int TW, int R, int CB, int DEPTH, bool NT, int ORDER>
__global__ __launch_bounds__(256) void strip_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int chunk_rows) {
    constexpr int CGS = CB / 4, LANES = 256 / CGS, ITEMS = R * TW / LANES;
    static_assert(ITEMS >= 1 && R * TW % LANES == 0, "shape");
    const unsigned cblocks = C / CB, strips = W / TW, chunks = H / chunk_rows;
    unsigned b = ORDER == 2 ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    unsigned cb, sx;
    if (ORDER == 1) { sx = b % strips; b /= strips; cb = b % cblocks; b /= cblocks; }
    else { cb = b % cblocks; b /= cblocks; sx = b % strips; b /= strips; }
    const unsigned cy = b % chunks;
    const long n = b / chunks;
    const int cg = threadIdx.x % CGS, lane = threadIdx.x / CGS;
    unsigned off[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int p = lane + LANES * i, row = p / TW, px = p % TW;
        off[i] = ((unsigned)(row * W + px) * C + cb * CB + cg * 4) * 4u;
    }
    const long pix0 = (n * H + (long)cy * chunk_rows) * W + sx * TW;
    const char* src = (const char*)(x + pix0 * C);
    char* dst = (char*)(y + pix0 * C);
    const long step = (long)R * W * C * 4;
    const int nsteps = chunk_rows / R;
    f32x4 v[DEPTH][ITEMS];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (d < nsteps) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) v[d][i] = NT ? __builtin_nontemporal_load((const f32x4*)(src + off[i])) : *(const f32x4*)(src + off[i]);
        }
        src += step;
    }
    for (int s = 0; s < nsteps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s + d >= nsteps) break;
            f32x4 r[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { r[i] = v[d][i]; r[i] = r[i] > 0.f ? r[i] : r[i] * 0.3f; }
            if (s + d + DEPTH < nsteps) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) v[d][i] = NT ? __builtin_nontemporal_load((const f32x4*)(src + off[i])) : *(const f32x4*)(src + off[i]);
            }
            src += step;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { if (NT) __builtin_nontemporal_store(r[i], (f32x4*)(dst + off[i])); else *(f32x4*)(dst + off[i]) = r[i]; }
            dst += step;
        }
    }
}

template <int TW, int R, int CB, int DEPTH, bool NT, int ORDER>
static void run(const char* name, const float* x, float* y, int N, int H, int W, int C, int chunk_rows) {
    const unsigned grid = (unsigned)((long)N * (H / chunk_rows) * (W / TW) * (C / CB));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = strip_kernel<TW, R, CB, DEPTH, NT, ORDER>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, x, y, N, H, W, C, chunk_rows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, x, y, N, H, W, C, chunk_rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-58s chunk %3d grid %6u : %.3f ms  %.2f TB/s\n", name, chunk_rows, grid, ms, 2.0 * N * H * W * C * 4 / ms / 1e9);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int N = 32, H = 256, W = 256, C = 384;
    const long n = (long)N * H * W * C;
    float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
    hipMemset(x, 0x3f, n * 4);
    for (int w = 0; w < 2; ++w) run<16, 8, 32, 1, false, 0>("warm", x, y, N, H, W, C, 256);
    for (int ch : {256, 64}) {
        run<16, 8, 32, 1, false, 0>("TW16 R8 CB32  depth1  (library pattern)", x, y, N, H, W, C, ch);
        run<16, 8, 32, 1, true, 0>("TW16 R8 CB32  depth1 nt", x, y, N, H, W, C, ch);
        run<16, 8, 32, 2, false, 0>("TW16 R8 CB32  depth2", x, y, N, H, W, C, ch);
        run<16, 8, 32, 2, true, 0>("TW16 R8 CB32  depth2 nt", x, y, N, H, W, C, ch);
        run<16, 8, 32, 1, false, 1>("TW16 R8 CB32  depth1 strip-fastest order", x, y, N, H, W, C, ch);
        run<16, 8, 32, 1, false, 2>("TW16 R8 CB32  depth1 no xcd remap", x, y, N, H, W, C, ch);
        run<16, 4, 64, 1, false, 0>("TW16 R4 CB64  depth1", x, y, N, H, W, C, ch);
        run<16, 8, 64, 1, false, 0>("TW16 R8 CB64  depth1 (8 items)", x, y, N, H, W, C, ch);
        run<16, 4, 64, 2, true, 0>("TW16 R4 CB64  depth2 nt", x, y, N, H, W, C, ch);
        run<16, 2, 128, 1, false, 0>("TW16 R2 CB128 depth1", x, y, N, H, W, C, ch);
        run<16, 4, 128, 1, false, 0>("TW16 R4 CB128 depth1 (8 items)", x, y, N, H, W, C, ch);
        run<16, 2, 128, 2, true, 0>("TW16 R2 CB128 depth2 nt", x, y, N, H, W, C, ch);
        run<8, 8, 64, 1, false, 0>("TW8  R8 CB64  depth1", x, y, N, H, W, C, ch);
        run<8, 8, 128, 1, false, 0>("TW8  R8 CB128 depth1 (8 items)", x, y, N, H, W, C, ch);
        run<32, 8, 32, 1, false, 0>("TW32 R8 CB32  depth1 (8 items)", x, y, N, H, W, C, ch);
        run<32, 4, 32, 1, false, 0>("TW32 R4 CB32  depth1", x, y, N, H, W, C, ch);
    }
    return 0;
}
