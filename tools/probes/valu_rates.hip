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
__global__ __launch_bounds__(768, 3) void probe(float* __restrict__ out, int iters, int producers, int prio_valu, int prio_mfma, int mfma_waves, int dep) {
    __shared__ volatile int done;
    const int tid = threadIdx.x;
    if (tid == 0) done = 0;
    __syncthreads();
    if (tid < 512) {
        if (BUSY && (tid >> 6) < mfma_waves) {
            if (prio_mfma == 1) __builtin_amdgcn_s_setprio(1);
            if (prio_mfma == 3) __builtin_amdgcn_s_setprio(3);
            bf16x8 a[4], b[3];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)((tid + i + j) & 7);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)((tid * 3 + i + j) & 3);
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            long n = 0;
            if (dep == 0)
                while (done < producers) {
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[q % 3], acc[t], 0, 0, 0);
                    ++n;
                }
            else if (dep == 1)       // one dependent chain: the wave is not ready while its previous MFMA is in flight
                while (done < producers) {
#pragma unroll
                    for (int q = 0; q < 24; ++q) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 3], b[q % 3], acc[0], 0, 0, 0);
                    ++n;
                }
            else                     // independent accumulators, the wave steps aside for 4 x (dep - 1) cycles after every MFMA
                while (done < producers) {
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[q % 3], acc[t], 0, 0, 0);
                            if (dep == 2) asm volatile("s_nop 3");
                            if (dep == 3) asm volatile("s_nop 7");
                            if (dep == 4) { asm volatile("s_nop 7\n s_nop 7"); }
                        }
                    ++n;
                }
            float s = 0.f;
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
            out[(size_t)blockIdx.x * 768 + tid] = s;
            if ((tid & 63) == 0) out[(size_t)256 * 768 + blockIdx.x * 8 + (tid >> 6)] = (float)n;   // MFMA k-steps this wave got through
        }
    } else if ((tid - 512) >> 6 < producers) {
        if (prio_valu == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_valu == 3) __builtin_amdgcn_s_setprio(3);
        unsigned r[8]; f32x2 p[8];
        for (int i = 0; i < 8; ++i) { r[i] = 0x3f800000u + tid * 8 + i; p[i][0] = 1.0f + i; p[i][1] = 0.5f * tid; }
        for (int it = 0; it < iters; ++it) burst<KIND>(r, p, 0x3f000000u + it);
        unsigned s = 0;
        for (int i = 0; i < 8; ++i) s += r[i] + (unsigned)p[i][0] + (unsigned)p[i][1];
        out[(size_t)blockIdx.x * 768 + tid] = (float)s;
        if ((tid & 63) == 0) atomicAdd((int*)&done, 1);
    }
}

static int g_prio_valu = 0, g_prio_mfma = 0, g_mfma_waves = 8, g_dep = 0;

template <int KIND, int BUSY>
static void run(float* out, int iters, const char* name, int producers) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(out + (size_t)256 * 768, 0, 4096 * 4);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, BUSY>), dim3(256), dim3(768), 0, 0, out, iters, producers, g_prio_valu, g_prio_mfma, g_mfma_waves, g_dep);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    float steps = 0;
    if (BUSY) { float h[8]; (void)hipMemcpy(h, out + (size_t)256 * 768, sizeof(h), hipMemcpyDeviceToHost); for (int i = 0; i < 8; ++i) steps += h[i]; }
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64);
    // MFMA share: k-steps x 24 MFMAs x 32 cycles over 4 SIMDs (block 0), relative to the elapsed cycles
    const double mfma_util = BUSY ? steps * 24 * 32 / 4 / (ms * 1e-3 * 2.4e9) : 0.0;
    if (BUSY) printf("  %-20s next to MFMA (%d waves, prio valu %d mfma %d, chain mode %d) %9.3f ms %9.2f cycles/instr/SIMD   MFMA pipe busy %.2f\n",
                     name, g_mfma_waves, g_prio_valu, g_prio_mfma, g_dep, ms, cyc, mfma_util);
    else printf("  %-20s alone %9.3f ms %9.2f cycles/instr/SIMD\n", name, ms, cyc);
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int producers = 4;
    float* out;
    if (hipMalloc(&out, (256 * 768 + 4096) * 4) != hipSuccess) return 1;
    printf("%d x 64 instructions per VALU wave, one VALU wave per SIMD, 256 blocks\n", iters);
    run<11, 0>(out, iters, "warm-up", producers);
#define ALONE(K, NAME) run<K, 0>(out, iters, NAME, producers);
    ALONE(11, "v_mov_b32") ALONE(0, "v_and_b32") ALONE(4, "v_lshlrev_b32") ALONE(1, "v_sub_f32") ALONE(5, "v_fma_f32") ALONE(8, "v_max_f32")
    ALONE(2, "v_cvt_pk_bf16_f32") ALONE(3, "v_perm_b32") ALONE(9, "v_and_or_b32") ALONE(6, "v_pk_add_f32") ALONE(7, "v_pk_fma_f32") ALONE(10, "v_pk_mul_f32")
    const int small = iters / 20 > 50 ? iters / 20 : 50;       // the starved configurations take ~10^4 cycles per instruction
    const int cfgs[][4] = {{0, 0, 8, 0}, {3, 0, 8, 0}, {0, 0, 4, 0}, {3, 0, 4, 0}, {0, 0, 8, 1}, {3, 0, 8, 1}, {0, 0, 4, 1}, {0, 0, 8, 2}, {0, 0, 8, 3}, {0, 0, 8, 4}, {3, 0, 8, 3}};
    for (auto& c : cfgs) {
        g_prio_valu = c[0]; g_prio_mfma = c[1]; g_mfma_waves = c[2]; g_dep = c[3];
        run<11, 1>(out, small, "v_mov_b32", producers);
        run<2, 1>(out, small, "v_cvt_pk_bf16_f32", producers);
        run<6, 1>(out, small, "v_pk_add_f32", producers);
    }
    return 0;
}
