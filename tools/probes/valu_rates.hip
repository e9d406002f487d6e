// Hardware probe (not part of the library): issue cost of the vector-ALU instructions the split-bf16 producer waves are made
// of, alone and next to a saturated MFMA stream on the same SIMDs.  768-thread blocks, one per CU: waves 8-11 (one per SIMD)
// run `iters` x 64 independent instructions of ONE kind; waves 0-7 either sleep (idle) or issue back-to-back
// v_mfma_f32_32x32x16_bf16 until the VALU waves are done.  Prints cycles per instruction per SIMD (2.4 GHz assumed).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__device__ __forceinline__ void burst(unsigned (&r)[8], f32x2 (&p)[8], unsigned k) {
    // 8 independent chains, 8 instructions each per call
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (KIND == 0) { asm volatile("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n"
                                      "v_and_b32 %4, 0xffff0000, %4\n v_and_b32 %5, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %6\n v_and_b32 %7, 0xffff0000, %7"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])); }
        if (KIND == 1) { asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                                      "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k)); }
        if (KIND == 2) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n"
                                      "v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k)); }
        if (KIND == 3) { asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                                      "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k), "v"(0x07060302u)); }
        if (KIND == 4) { asm volatile("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n"
                                      "v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])); }
        if (KIND == 5) { asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                                      "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k)); }
        if (KIND == 6) { asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                      "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                                      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[0])); }
        if (KIND == 7) { asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                                      "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                                      : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[1])); }
        if (KIND == 8) { asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                                      "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k)); }
        if (KIND == 9) { asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n"
                                      "v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9"
                                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k), "v"(0x1u)); }
        if (KIND == 10) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                       "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                       : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(p[1])); }
        if (KIND == 11) { asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                                       "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(k)); }
    }
}

template <int KIND, int BUSY>
__global__ __launch_bounds__(768, 3) void probe(float* __restrict__ out, int iters, int producers) {
    __shared__ volatile int done;
    const int tid = threadIdx.x;
    if (tid == 0) done = 0;
    __syncthreads();
    if (tid < 512) {
        if (BUSY) {
            bf16x8 a[4], b[3];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)((tid + i + j) & 7);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)((tid * 3 + i + j) & 3);
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            long n = 0;
            while (done < producers) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[q % 3], acc[t], 0, 0, 0);
                ++n;
            }
            float s = 0.f;
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
            out[(size_t)blockIdx.x * 768 + tid] = s;
            if ((tid & 63) == 0) out[(size_t)256 * 768 + blockIdx.x * 8 + (tid >> 6)] = (float)n;   // MFMA k-steps this wave got through
        }
    } else if ((tid - 512) >> 6 < producers) {
        unsigned r[8]; f32x2 p[8];
        for (int i = 0; i < 8; ++i) { r[i] = 0x3f800000u + tid * 8 + i; p[i][0] = 1.0f + i; p[i][1] = 0.5f * tid; }
        for (int it = 0; it < iters; ++it) burst<KIND>(r, p, 0x3f000000u + it);
        unsigned s = 0;
        for (int i = 0; i < 8; ++i) s += r[i] + (unsigned)p[i][0] + (unsigned)p[i][1];
        out[(size_t)blockIdx.x * 768 + tid] = (float)s;
        if ((tid & 63) == 0) atomicAdd((int*)&done, 1);
    }
}

template <int KIND, int BUSY>
static void run(float* out, int iters, const char* name, int producers) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, BUSY>), dim3(256), dim3(768), 0, 0, out, iters, producers);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<KIND, BUSY>), dim3(256), dim3(768), 0, 0, out, iters, producers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    float steps = 0;
    if (BUSY) { float h[8]; hipMemcpy(h, out + (size_t)256 * 768, sizeof(h), hipMemcpyDeviceToHost); for (int i = 0; i < 8; ++i) steps += h[i]; }
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64);
    // MFMA share: 8 waves x 24 MFMAs x 32 cycles per k-step over 4 SIMDs, relative to the elapsed cycles
    const double mfma_util = BUSY ? steps * 24 * 32 / 4 / (ms * 1e-3 * 2.4e9) : 0.0;
    printf("  %-22s %s  %8.3f ms  %6.2f cycles/instr/SIMD%s", name, BUSY ? "next to MFMA" : "alone       ", ms, cyc, BUSY ? "" : "\n");
    if (BUSY) printf("   MFMA pipe busy %.2f\n", mfma_util);
}

#define BOTH(K, NAME) run<K, 0>(out, iters, NAME, producers); run<K, 1>(out, iters, NAME, producers);
int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int producers = argc > 2 ? atoi(argv[2]) : 4;
    float* out;
    if (hipMalloc(&out, (256 * 768 + 4096) * 4) != hipSuccess) return 1;
    printf("%d x 64 instructions per VALU wave, %d VALU wave(s) per CU (one per SIMD), 256 blocks\n", iters, producers);
    BOTH(11, "v_mov_b32") BOTH(0, "v_and_b32") BOTH(4, "v_lshlrev_b32") BOTH(1, "v_sub_f32") BOTH(5, "v_fma_f32") BOTH(8, "v_max_f32")
    BOTH(2, "v_cvt_pk_bf16_f32") BOTH(3, "v_perm_b32") BOTH(9, "v_and_or_b32") BOTH(6, "v_pk_add_f32") BOTH(7, "v_pk_fma_f32") BOTH(10, "v_pk_mul_f32")
    return 0;
}
