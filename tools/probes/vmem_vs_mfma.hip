// Hardware probe (not part of the library): how much does one vector-memory instruction issued by a co-resident wave
// cost the MFMA stream of its SIMD?  768-thread blocks, one per CU: waves 0-7 run a pure v_mfma_f32_32x32x16_bf16
// loop (accumulators in VGPRs, like gemm_pc.hip), waves 8-11 issue `per` loads per 96 MFMAs-worth of time in one of
// several forms.  Prints ms per variant.   hipcc --offload-arch=gfx950 -O3 -o vmem_vs_mfma vmem_vs_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(768, 3) void probe(const float* __restrict__ src, float* __restrict__ out, int iters, int per, int sleep, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x;
    if (tid < 512) {
        bf16x8 a[4], b[3];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)((tid + i + j) & 7);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)((tid * 3 + i + j) & 3);
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[q % 3], acc[t], 0, 0, 0);
            if (FORM == 6 && (it & 1) == 0) {       // loads issued by the MFMA waves themselves (3 per 48 MFMAs)
                const f32x4* p = reinterpret_cast<const f32x4*>(src) + ((size_t)blockIdx.x * 4096 + (it & 63) * 64 + (tid & 63));
                f32x4 v0 = p[0], v1 = p[64 * 64], v2 = p[128 * 64];
                b[0][0] = (__bf16)(v0[0] + v1[1] + v2[2]);
            }
        }
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[(size_t)blockIdx.x * 768 + tid] = s;
    } else {
        const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane((tid - 512) >> 6);
        const char* base = reinterpret_cast<const char*>(src) + ((size_t)blockIdx.x * 4 + w) * (1 << 20);
        unsigned acc = 0;
        // iters k-steps of 24 MFMAs per consumer wave = 768 pipe cycles per SIMD per step (2 waves): one "stage" = 2 steps
        const int stages = iters / 2;
        for (int s = 0; s < stages; ++s) {
            const unsigned off = (unsigned)(((s & 127) * 64 + lane) * 16);
            for (int j = 0; j < per; ++j) {
                if (FORM == 1) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off + j * 8192), "s"(base) : "memory"); asm volatile("" :: "v"(v)); }
                if (FORM == 2) { f32x4 v; const char* p = base + off + j * 8192; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); asm volatile("" :: "v"(v)); }
                if (FORM == 3) { float v; asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"((unsigned)(lane * 4 + j * 8192)), "s"(base) : "memory"); asm volatile("" :: "v"(v)); }
                if (FORM == 4) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned*)(base + off + j * 8192), (__attribute__((address_space(3))) unsigned*)(lds + w * 16384 + j * 1024), 16, 0, 0); }
                if (FORM == 5) { f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane * 16 + j * 1024 + s * 64) & 65535 & ~15)); acc += (unsigned)v[0]; }   // LDS reads instead
                if (FORM == 7) { f32x4 v = {1.f, 2.f, 3.f, (float)s}; *reinterpret_cast<f32x4*>(lds + ((lane * 16 + j * 1024 + w * 16384) & 65535)) = v; }     // LDS writes
                if (FORM == 8) { acc = acc * 1664525u + 1013904223u; acc ^= acc >> 7; acc += lane; acc = acc * 3u + 1u; }                      // VALU only
            }
            for (int z = 0; z < sleep; ++z) __builtin_amdgcn_s_sleep(8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (acc == 0x12345678u) sink[0] = acc;
    }
}

template <int FORM>
static float run(const float* src, float* out, unsigned* sink, int iters, int per, int sleep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<FORM>, dim3(256), dim3(768), 0, 0, src, out, iters, per, sleep, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe<FORM>, dim3(256), dim3(768), 0, 0, src, out, iters, per, sleep, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int iters = 4096;
    float* src; float* out; unsigned* sink;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
    CK(hipMalloc(&src, (size_t)1 << 31)); CK(hipMemset(src, 0, (size_t)1 << 31));
    CK(hipMalloc(&out, 256 * 768 * 4)); CK(hipMalloc(&sink, 64));
    printf("src %p out %p sink %p\n", (void*)src, (void*)out, (void*)sink);
    const double mfma_cycles = (double)iters * 24 * 2 * 32;     // per SIMD
    printf("ideal MFMA-bound: %.3f ms at 2.0 GHz\n", mfma_cycles / 2.0e6);
    for (int warm = 0; warm < 3; ++warm) { run<0>(src, out, sink, iters, 0, 12); CK(hipDeviceSynchronize()); CK(hipGetLastError()); }
    printf("warm ok\n");
    const int sl = argc > 1 ? atoi(argv[1]) : 4;      // s_sleep 8 (~512 clk) repeats per stage
    printf("form 0 (idle loader waves, sleep only)          : %.3f ms\n", run<0>(src, out, sink, iters, 0, sl));
    for (int per = 4; per <= 16; per *= 2) {
        printf("per=%2d loads / stage / loader wave\n", per);
        printf("  form 1 global_load_dwordx4 saddr+voffset     : %.3f ms\n", run<1>(src, out, sink, iters, per, sl));
        printf("  form 3 global_load_dword                      : %.3f ms\n", run<3>(src, out, sink, iters, per, sl));
        printf("  form 4 global_load_lds 16 B (LDS-DMA)         : %.3f ms\n", run<4>(src, out, sink, iters, per, sl));
        printf("  form 5 ds_read_b128 (LDS reads)               : %.3f ms\n", run<5>(src, out, sink, iters, per, sl));
        printf("  form 7 ds_write_b128 (LDS writes)             : %.3f ms\n", run<7>(src, out, sink, iters, per, sl));
        printf("  form 8 VALU only (5 ops)                      : %.3f ms\n", run<8>(src, out, sink, iters, per, sl));
    }
    printf("form 6 (3 loads per 2 k-steps by the MFMA waves)  : %.3f ms\n", run<6>(src, out, sink, iters, 0, sl));
    return 0;
}
