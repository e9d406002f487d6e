// bf16 ACTIVATION STORAGE (BASELINE config 5: "mixed bf16"): helpers shared by bf16_gemm.hip / bf16_dw.hip / bf16_elem.hip.
//
// Contract of the tsii_bf16_* entry points (include/tsii_hip.h): activations and activation gradients live in HBM as bf16
// NHWC ([N,H,W,C], C % 8 == 0: one 16-byte vector = 8 channels); parameters, parameter gradients, BatchNorm statistics and
// every accumulation are fp32.  A kernel reads bf16, computes in fp32 and rounds ONCE (RNE, v_cvt_pk_bf16_f32) when it
// stores; statistics a kernel emits about its own output are taken over the ROUNDED values (the consumer normalises what
// is in memory, not what was in the accumulator).  The segmentation nets have no mask planes, so these kernels carry none.
#pragma once
#include "tsii_common.h"

namespace tsii {

typedef unsigned short bf16_t;      // raw bits
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hu32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
typedef float hf32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// two fp32 -> one dword of two bf16 (element 0 in the low half), RNE: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {
    const hf32x2 x = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, hbf16x2));
}
__device__ __forceinline__ float bf16_round(float a) { return bf16_lo(bf16_pack2(a, 0.f)); }
__device__ __forceinline__ bf16_t bf16_bits(float a) { return (bf16_t)(bf16_pack2(a, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack8(const hu32x4 u, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf16_lo(u[j]); v[2 * j + 1] = bf16_hi(u[j]); }
}
__device__ __forceinline__ hu32x4 pack8(const float (&v)[8]) {
    hu32x4 u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = bf16_pack2(v[2 * j], v[2 * j + 1]);
    return u;
}
// 8 consecutive bf16 (16-byte aligned)
__device__ __forceinline__ hu32x4 ld8(const bf16_t* __restrict__ p) { return *reinterpret_cast<const hu32x4*>(p); }
__device__ __forceinline__ hu32x4 ld8_nt(const bf16_t* __restrict__ p) { return __builtin_nontemporal_load(reinterpret_cast<const hu32x4*>(p)); }
__device__ __forceinline__ void st8(bf16_t* __restrict__ p, const hu32x4 u) { *reinterpret_cast<hu32x4*>(p) = u; }
__device__ __forceinline__ void st8_nt(bf16_t* __restrict__ p, const hu32x4 u) { __builtin_nontemporal_store(u, reinterpret_cast<hu32x4*>(p)); }

// ds_read_b64_tr_b16 (gfx950): 8 bytes per lane at the lane's own address, delivered TRANSPOSED inside each 16-lane group -- lane q of a
// group gets element (q & 3) of the 8 bytes lanes 4 i + (q >> 2), i = 0 .. 3, of its group supplied (measured, tools/probes/tr_read.hip).
// With lane q' pointing at (row q' >> 2, columns 4 (q' & 3) ..) of a [4 rows][16 columns] block of a row-major bf16 image, lane q gets
// column q of the four rows: four consecutive k of one channel, i.e. half an MFMA operand of an operand stored [k][channel].
#ifndef TSII_ASYNC_LOADS
template <int OFF> __device__ __forceinline__ void lds_read8_tr(hu32x2& d, const void* p) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))), "n"(OFF) : "memory");
}
template <int N, class A, class B, class C, class D>
__device__ __forceinline__ void lds_wait4(A& a, B& b, C& c, D& d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory"); }
#endif

// load-time BatchNorm + activation (tsii_common.h: bn_act_load) on 8 channels; sc == nullptr at the call site means "plain"
struct InBN8 {
    float sc[8], sh[8];
};
__device__ __forceinline__ void load_inbn8(const InBN& ib, int c, InBN8& o) {
    const VecF<4> a = vload<4>(ib.sc + c), b = vload<4>(ib.sc + c + 4), d = vload<4>(ib.sh + c), e = vload<4>(ib.sh + c + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.sc[i] = a.v[i]; o.sc[4 + i] = b.v[i]; o.sh[i] = d.v[i]; o.sh[4 + i] = e.v[i]; }
}
__device__ __forceinline__ void apply_inbn8(float (&v)[8], const InBN8& c, float neg, float hi) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bn_act_load(v[i], c.sc[i], c.sh[i], neg, hi);
}

// derivative of the load-time activation min(max(z, neg z), hi) at pre-activation z (torch semantics, like act_grad)
__device__ __forceinline__ float inbn_grad(float z, float neg, float hi) { return (z > 0.f && z < hi) ? 1.f : (z > 0.f ? 0.f : neg); }

// out[n] = sum_m a[m, n] of a bf16 [M, N] matrix (N % 8 == 0); ws: partial_rows(M, N / 8) * N floats   (bf16_gemm.hip)
int launch_bf16_colsum(const bf16_t* a, int64_t M, int N, float* out, float* ws, hipStream_t st);

}  // namespace tsii
