// Hardware probe (not part of the library): how many independent vector-ALU instructions fit in the shadow of one
// v_mfma_f32_32x32x16_bf16 when THE SAME WAVE issues both (software-interleaved, the alternative to separate producer waves)?
// Every wave of a block runs: 24 x { MFMA ; NV x VALU } per step, accumulators and VALU chains independent.  Waves per SIMD = 1, 2, 3.
// Prints cycles per MFMA per SIMD (2.4 GHz assumed; 32 = the matrix pipe's own rate).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__device__ __forceinline__ void valu(unsigned (&r)[8], unsigned k, unsigned char* lds, int lane, int q) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = (q * NV + j) & 7;
        if (KIND == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[i]));
        if (KIND == 1) { if ((j & 3) == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                         else if ((j & 3) == 1) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[i]));
                         else if ((j & 3) == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                         else asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[i])); }
    }
    if (KIND == 2 && (q & 1) == 0) {        // + one ds_write_b128 per two MFMAs
        f32x4 v = {1.f, 2.f, 3.f, (float)q};
        *reinterpret_cast<volatile f32x4*>(lds + ((lane * 16 + q * 1024) & 32767)) = v;
    }
    if (KIND == 3 && (q & 1) == 0) {        // + one ds_read_b128 per two MFMAs
        f32x4 v = *reinterpret_cast<volatile f32x4*>(lds + ((lane * 16 + q * 1024) & 32767));
        r[q & 7] += (unsigned)v[0];
    }
}

template <int NV, int KIND>
__global__ __launch_bounds__(768, 3) void probe(float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a[4], b[3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)((tid + i + j) & 7);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)((tid * 3 + i + j) & 3);
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = 0x3f800000u + tid * 8 + i;
    unsigned char* mylds = lds + (tid >> 6 & 1) * 32768;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 3], b[q % 3], acc[q & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu<(KIND >= 2 ? 4 : NV), (KIND >= 2 ? 1 : KIND)>(r, 0x3f000000u + it, mylds, lane, q);
            if (KIND >= 2) valu<0, KIND>(r, 0, mylds, lane, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int k = 0; k < 16; ++k) s += acc[t][k];
    for (int i = 0; i < 8; ++i) s += (float)r[i];
    out[(size_t)blockIdx.x * 768 + tid] = s;
}

template <int NV, int KIND>
static void run(float* out, int iters, int waves, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(waves * 64), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(waves * 64), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 24 * (waves / 4.0);      // MFMAs per SIMD
    printf("  %2d waves/CU  %-34s %8.3f ms  %6.1f cycles per MFMA per SIMD  (matrix pipe busy %.2f)\n", waves, what, ms,
           ms * 1e-3 * 2.4e9 / per_simd, 32.0 * per_simd / (ms * 1e-3 * 2.4e9));
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* out;
    if (hipMalloc(&out, 256 * 768 * 4) != hipSuccess) return 1;
    for (int waves = 4; waves <= 12; waves += 4) {
        run<0, 0>(out, iters, waves, "MFMA only");
        run<1, 1>(out, iters, waves, "+1 VALU per MFMA (split mix)");
        run<2, 1>(out, iters, waves, "+2 VALU per MFMA (split mix)");
        run<3, 1>(out, iters, waves, "+3 VALU per MFMA (split mix)");
        run<4, 1>(out, iters, waves, "+4 VALU per MFMA (split mix)");
        run<6, 1>(out, iters, waves, "+6 VALU per MFMA (split mix)");
        run<8, 1>(out, iters, waves, "+8 VALU per MFMA (split mix)");
        run<4, 0>(out, iters, waves, "+4 v_and_b32 per MFMA");
        run<4, 2>(out, iters, waves, "+4 VALU + ds_write_b128 / 2 MFMA");
        run<4, 3>(out, iters, waves, "+4 VALU + ds_read_b128 / 2 MFMA");
    }
    return 0;
}
