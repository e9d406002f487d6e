// Pieces shared by the split-bf16 GEMM kernels (gemm_split.hip: 4-wave tiles; gemm_pc.hip: producer / consumer form).
#pragma once
#include "gemm_tiles.h"

namespace tsii {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int SPLIT_BK = 32;          // K elements per tile; one LDS row = 32 bf16 = 64 bytes = 4 chunks of 8

// a - b as ONE v_sub_f32 the vectorizer cannot pair into a packed-fp32 instruction
__device__ __forceinline__ float scalar_sub(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a - b;
#endif
}

// 8 consecutive fp32 values -> P planes of 8 bf16 (element i of a plane in bits [16*(i&1), +16) of dword i >> 1)
template <int P>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&pl)[P]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 x = {v[2 * j], v[2 * j + 1]};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bf16x2 h = __builtin_convertvector(x, bf16x2);      // v_cvt_pk_bf16_f32 (RNE)
            const unsigned u = __builtin_bit_cast(unsigned, h);
            pl[p][j] = u;
            if (p + 1 < P) {
                // exact: the low significand bits.  Two scalar v_sub_f32 ON PURPOSE: the packed form (v_pk_add_f32) issues at
                // ~1/15 of the scalar rate while the SIMD's matrix pipe is busy (tools/probes/valu_rates.hip, measured on MI355X:
                // 9 vs 100-180 cycles per instruction next to a v_mfma_f32_32x32x16_bf16 stream)
                x[0] = scalar_sub(x[0], __builtin_bit_cast(float, u << 16));
                x[1] = scalar_sub(x[1], __builtin_bit_cast(float, u & 0xffff0000u));
            }
        }
    }
}

// byte offset of (row, chunk) inside one plane of an operand tile
__device__ __forceinline__ int split_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// PRODUCTS = 8 / 6 -> 3 planes, 3 -> 2 planes.  Order of the partial products (smallest first):
//   8: (2,1) (1,2) (2,0) (1,1) (0,2) (1,0) (0,1) (0,0)   6: the last six of those   3: (1,0) (0,1) (0,0)   1: (0,0) = plain bf16 operands
template <int PRODUCTS>
struct SplitTerm {
    static __device__ __forceinline__ constexpr int pa(int q) {
        if (PRODUCTS == 1) return 0;
        if (PRODUCTS == 3) return q == 0 ? 1 : 0;
        const int qq = q + (8 - PRODUCTS);      // index into the 8-term order
        return qq == 0 ? 2 : qq == 1 ? 1 : qq == 2 ? 2 : qq == 3 ? 1 : qq == 4 ? 0 : qq == 5 ? 1 : 0;
    }
    static __device__ __forceinline__ constexpr int pb(int q) {
        if (PRODUCTS == 1) return 0;
        if (PRODUCTS == 3) return q == 1 ? 1 : 0;
        const int qq = q + (8 - PRODUCTS);
        return qq == 0 ? 1 : qq == 1 ? 2 : qq == 2 ? 0 : qq == 3 ? 1 : qq == 4 ? 2 : qq == 5 ? 0 : qq == 6 ? 1 : 0;
    }
};
template <int PRODUCTS>
struct SplitPlanes { static constexpr int value = PRODUCTS == 1 ? 1 : PRODUCTS == 3 ? 2 : 3; };

}  // namespace tsii
