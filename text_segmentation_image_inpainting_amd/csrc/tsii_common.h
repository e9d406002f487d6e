// Shared helpers for the gfx950 kernels of libtsii_hip.so (see include/tsii_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/tsii_hip.h"

namespace tsii {

// thread-local error string behind tsii_last_error()
void set_error(const char* fmt, ...);
// hipGetLastError() after a launch -> 0 / negative code with message
int check_launch(const char* what);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Streaming kernels use a capped grid + grid-stride loop: 256 CUs x 8 blocks.
static inline unsigned stream_grid(int64_t work_items, int block) {
    int64_t g = cdiv64(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Element-wise streaming kernels: ONE vector per thread, as many blocks as that takes.  Measured on the MI355X
// (tools/probes/stream_copy.hip, 3.2 GB in + 3.2 GB out): a capped grid of 2048 blocks walking the tensor with a
// grid-stride loop copies at 4.7 TB/s whatever the unroll depth, one float4 per thread at 6.1 TB/s, 6.5 TB/s with
// non-temporal loads and stores (the dispatcher's block order sweeps the address range once, front to back).  The
// kernels keep their grid-stride loop (any grid is correct); this grid makes it run once.
static inline unsigned flat_grid(int64_t work_items, int block) {
    int64_t g = cdiv64(work_items, block);
    if (g > 0x7fffffffll) g = 0x7fffffffll;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == TSII_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TSII_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == TSII_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    if (act == TSII_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
// derivative as a function of the pre-activation value (torch semantics: 0 at v <= 0,
// ReLU6 passes gradient strictly inside (0,6))
__device__ __forceinline__ float act_grad(float v, int act, float slope) {
    if (act == TSII_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == TSII_ACT_LEAKY) return v > 0.f ? 1.f : slope;
    if (act == TSII_ACT_RELU6) return (v > 0.f && v < 6.f) ? 1.f : 0.f;
    if (act == TSII_ACT_SIGMOID) { const float sg = 1.f / (1.f + expf(-v)); return sg * (1.f - sg); }
    return 1.f;
}

// K6b: BatchNorm(+ReLU-family activation) of the producer applied to an operand while it is loaded:
//   z = sc[c]*v + sh[c];  a = min(max(z, neg*z), hi)      (4 VALU ops, branch-free, NaN propagates)
//   NONE: neg 1, hi inf | RELU: 0, inf | LEAKY: slope in [0,1], inf | RELU6: 0, 6.   sc == nullptr: no transform.
struct InBN {
    const float* sc;
    const float* sh;
    float neg;
    float hi;
};
__device__ __forceinline__ float bn_act_load(float v, float sc, float sh, float neg, float hi) {
    const float z = fmaf(v, sc, sh);
    return fminf(fmaxf(z, neg * z), hi);
}
static inline int make_in_bn(const float* sc, const float* sh, int act, float slope, InBN* out) {
    out->sc = sc; out->sh = sh; out->hi = __builtin_huge_valf();
    if (act == TSII_ACT_NONE) out->neg = 1.f;
    else if (act == TSII_ACT_RELU) out->neg = 0.f;
    else if (act == TSII_ACT_LEAKY && slope >= 0.f && slope <= 1.f) out->neg = slope;
    else if (act == TSII_ACT_RELU6) { out->neg = 0.f; out->hi = 6.f; }
    else return -1;   // sigmoid, leaky slope outside [0,1]: no load-time form
    return 0;
}

// row scale (include/tsii_hip.h): r0 == nullptr -> 1
struct RowScale {
    const float* r0;
    const float* r1;
    int split;
};
__device__ __forceinline__ float row_scale_at(const RowScale& rs, int64_t row, int ch) {
    if (rs.r0 == nullptr) return 1.f;
    return ch < rs.split ? rs.r0[row] : (rs.r1 != nullptr ? rs.r1[row] : 1.f);
}

// generic helpers shared across translation units (reduce.hip)
// out[j] = sum_r ws[r*len + j]  (double accumulation), r in [0,rows)
int launch_reduce_rows(const float* ws, int rows, int64_t len, float* out, hipStream_t stream);
// the same over partial rows laid out [co][t][ci], written as the weight gradient [co][ci][t]
int launch_reduce_rows_conv(const float* ws, int rows, int cout, int cin, int T, float* out, hipStream_t stream);
// out[c*rows_in + r] = in[r*cols_in + c]  (2-D transpose of a [rows_in, cols_in] matrix)
int launch_transpose(const float* in, int rows_in, int cols_in, float* out, hipStream_t stream);

// BatchNorm backward: partial rows [rows][2][c] of (sum dz, sum dz*xhat) -> dgamma, dbeta and the apply pass's table of
// per-channel constants coef[6][c] = (mean, 1/std, gamma, beta, sum dz / m, sum dz*xhat / m)   (bn.hip)
size_t bn_bwd_reduce_ws_bytes(int64_t rows, int c);
int launch_bn_bwd_reduce(const float* mean, const float* var, const float* gamma, const float* beta, float eps, int training,
                         const float* bwd_part, int64_t rows, int64_t m, int c, float* dgamma, float* dbeta, void* l1_ws, float* coef, hipStream_t st);

// out[n] = sum_m a[m*N+n] * rowmul[m] (rowmul may be null); ws: colsum_ws_floats(M,N) floats
size_t colsum_ws_floats(int64_t M, int N);
int launch_colsum_scaled(const float* a, const float* rowmul, int64_t M, int N, float* out, float* ws,
                         hipStream_t stream);

// dense convolution as implicit GEMM on the MFMA kernels (gemm.hip); weights must be pre-laid-out:
//   forward / dW: wr[co][t*cin + ci];   dX: wd[ci][t*cout + co]
struct ConvGemmGeom {
    int n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, ho, wo;
};
bool conv_gemm_ok(const ConvGemmGeom& g);        // vector gather: cin % 4 == 0, no per-channel mask
bool conv_gemm_elem_ok(const ConvGemmGeom& g);   // element-wise gather: few input channels / per-channel mask
int launch_conv_gemm_fwd(const float* x, const float* mfull, RowScale rs, const float* wr, const float* bias,
                         const float* denom, const float* keep, const ConvGemmGeom& g, float* y, hipStream_t st, float* stats = nullptr);
int launch_conv_gemm_dx(const float* dy, const float* inv, const float* wd, RowScale rs_out, const ConvGemmGeom& g,
                        float* dx, hipStream_t st);
bool conv_gemm_dx_phases_ok(const ConvGemmGeom& g);
int launch_conv_gemm_dx_phases(const float* dy, const float* inv, const float* w, float* wd_ws, RowScale rs_out,
                               const ConvGemmGeom& g, float* dx, hipStream_t st);
size_t conv_gemm_dw_ws_floats(const ConvGemmGeom& g);
int launch_conv_gemm_dw(const float* dy, const float* inv, const float* x, const float* mfull, RowScale rs,
                        const ConvGemmGeom& g, float* dwgt, float* ws, hipStream_t st);

// number of partial rows for per-channel reductions over M rows with CG channel groups
static inline int partial_rows(int64_t M, int CG) {
    int64_t r = 131072 / (CG > 0 ? CG : 1);
    if (r > 4096) r = 4096;
    if (r > M) r = M;
    if (r < 1) r = 1;
    return (int)r;
}

// Grid for flat streaming kernels whose threads keep a FIXED channel group across the grid-stride
// loop: (grid * block) is a multiple of CG, so (global thread id % CG) never changes and the
// per-channel constants live in registers (no per-element div/mod).
static inline unsigned chan_grid(int64_t items, int CG, int block) {
    int a = CG, b = block;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = CG / a;                 // grid must be a multiple of this
    int64_t want = cdiv64(items, block);
    if (want > 2048) want = 2048;
    int64_t k = want / unit;
    if (k < 1) k = 1;
    return (unsigned)(k * unit);
}

// XCD-aware bijective block remap.  The dispatcher deals consecutive block ids round-robin to the 8
// XCDs (private L2 each); this maps block id -> logical tile so that every XCD works on ONE contiguous
// range of logical tiles (neighbouring tiles share operand panels / stencil rows in that XCD's L2).
// Placement only affects speed, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned xcd = bid & 7u, local = bid >> 3;
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// W consecutive floats handled by one thread (W == 4: one 16-byte access)
template <int W>
struct VecF {
    float v[W];
};
template <int W>
__device__ __forceinline__ VecF<W> vload(const float* __restrict__ p) {
    VecF<W> r;
    if constexpr (W == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) r.v[i] = p[i];
    }
    return r;
}
// non-temporal forms for tensors that are streamed once (activations far larger than the caches)
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <int W>
__device__ __forceinline__ VecF<W> vload_nt(const float* __restrict__ p) {
    VecF<W> r;
    if constexpr (W == 4) {
        const f32x4s t = __builtin_nontemporal_load(reinterpret_cast<const f32x4s*>(p));
        r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) r.v[i] = __builtin_nontemporal_load(p + i);
    }
    return r;
}
template <int W>
__device__ __forceinline__ void vstore_nt(float* __restrict__ p, const VecF<W>& r) {
    if constexpr (W == 4) {
        const f32x4s t = {r.v[0], r.v[1], r.v[2], r.v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4s*>(p));
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) __builtin_nontemporal_store(r.v[i], p + i);
    }
}
template <int W>
__device__ __forceinline__ void vstore(float* __restrict__ p, const VecF<W>& r) {
    if constexpr (W == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) p[i] = r.v[i];
    }
}

}  // namespace tsii

// optimisation barrier on a 32-bit VGPR value: the compiler may not re-derive it algebraically (used to keep running
// address offsets running); the test emulator's hip_runtime.h supplies a host form
#ifndef TSII_OPAQUE_U32
#define TSII_OPAQUE_U32(x) asm volatile("" : "+v"(x))
#endif

// Counted asynchronous global loads.  hipcc's own s_waitcnt insertion is conservative across loop iterations: in a
// software-pipelined loop it waits for loads issued one iteration ago however far ahead they were requested (measured
// in gemm_pc.hip: the producer waves ran at one memory latency per stage).  These loads are invisible to that pass; the
// caller waits with an explicit count = the number of loads it issued after the one it needs, and passes the destination
// registers through the wait so that no use can be scheduled above it.  (tests/emu supplies synchronous host forms.)
#ifndef TSII_ASYNC_LOADS
namespace tsii {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void async_load16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void async_load16(u32x4v& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void async_load4(float& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// wave-uniform 64-bit base (SGPR pair) + 32-bit byte offset per lane: no address arithmetic on the vector pipe
__device__ __forceinline__ void async_load16(f32x4& d, const void* base, unsigned off) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void async_load4(float& d, const void* base, unsigned off) { asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory"); }
template <int N, class A> __device__ __forceinline__ void async_wait(A& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N, class A, class B> __device__ __forceinline__ void async_wait(A& a, B& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N, class A, class B, class C, class D>
__device__ __forceinline__ void async_wait(A& a, B& b, C& c, D& d) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory"); }
// block barrier for LDS traffic only: __syncthreads() also drains vmcnt, i.e. every load in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
// Direct global -> LDS load (global_load_lds_dwordx4): lane i's 16 bytes (per-lane global address) land at lds_wave_base + 16 i --
// the LDS side is wave-uniform base + lane-linear, swizzles go on the SOURCE address.  Counted by vmcnt like any other load:
// async_wait_lds<N>() waits until at most N younger loads of this wave are outstanding; a following lds_barrier() makes the
// other waves' pieces visible.  No staging registers: the prefetch depth is LDS stages, not VGPRs.
__device__ __forceinline__ void async_load16_lds(void* lds_wave_base, const void* gptr) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gptr)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base))), 16, 0, 0);
}
template <int N> __device__ __forceinline__ void async_wait_lds() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS read the compiler's scheduler leaves where it stands (it sinks a plain ds_read down to its first use, and will not lift one
// over a direct-to-LDS load anyway): 16 bytes at the LDS pointer p + OFF.  Counted by lgkmcnt, LDS operations return in order:
// lds_wait<N>(regs...) returns once at most N younger ones are outstanding and ties the registers to that point.
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4v& d, const void* p) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(static_cast<unsigned>(reinterpret_cast<uintptr_t>(p))), "n"(OFF) : "memory");
}
template <int N, class A, class B, class C, class D, class E, class F>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c, D& d, E& e, F& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}
}  // namespace tsii
#else
namespace tsii {
typedef f32x4_emu2 f32x4;
typedef u32x4_emu2 u32x4v;
}
#endif

#define TSII_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            tsii::set_error(__VA_ARGS__);       \
            return -1;                          \
        }                                       \
    } while (0)
