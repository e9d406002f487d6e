// TEST-ONLY host emulation of the small HIP subset the kernels in
// text_segmentation_image_inpainting_amd/csrc use.  It shadows <hip/hip_runtime.h>
// when tests/emu/build_emu.py compiles the UNMODIFIED product sources with the host
// clang++, so kernel index math, LDS tiling, wave shuffles and the MFMA fragment
// layout can be checked against the oracle in the CPU-only container.  It is never
// built into, loaded by, or reachable from the product library (which has no CPU
// path at all); only tests/ load the resulting tests/emu/_build/libtsii_emu.so.
//
// Model: one OS thread; every HIP thread of a block is a ucontext fiber; blocks run
// one after another.  __syncthreads() and the wave collectives (__shfl_*, MFMA) are
// cooperative yield points.  Wavefront = 64 lanes, MFMA 32x32x2 f32 fragment layout
// as documented for gfx950 (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]); MFMA 32x32x16 bf16: lane l holds A[i=l&31][k=8*(l>>5)+j] and
// B[k=8*(l>>5)+j][col=l&31], j = 0..7, same D map, fp32 accumulation in k order.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define TSII_HIP_EMU 1
#define TSII_OPAQUE_U32(x) asm volatile("" : "+r"(x))
#define TSII_PIN_F2(x) ((void)0)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef float f32x4_emu __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_emu __attribute__((ext_vector_type(8)));

namespace hipemu {
extern dim3 tIdx, bIdx, bDim, gDim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
void barrier_only();
void async_issue(void* dst, const void* src, int bytes);
void async_retire(int keep);
void yield();            // spin-wait loops (s_sleep): let the other fibers of the block run
unsigned exchange32(unsigned v, int src_lane_delta_mode, int arg, int width);
f32x16_emu mfma_32x32x2(float a, float b, f32x16_emu c);
f32x4_emu mfma_16x16x4(float a, float b, f32x4_emu c);
f32x16_emu mfma_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c);
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::tIdx)
#define blockIdx (hipemu::bIdx)
#define blockDim (hipemu::bDim)
#define gridDim (hipemu::gDim)
#define warpSize 64

static inline void __syncthreads() { hipemu::syncthreads(); }

template <class T> static inline T __emu_x(T v, int mode, int arg, int width) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    unsigned u; std::memcpy(&u, &v, 4);
    u = hipemu::exchange32(u, mode, arg, width);
    T r; std::memcpy(&r, &u, 4);
    return r;
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) { return __emu_x(v, 0, (int)delta, width); }
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) { return __emu_x(v, 1, (int)delta, width); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return __emu_x(v, 2, mask, width); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { return __emu_x(v, 3, src, width); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

// wave-uniform by construction wherever the kernels use it
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) { hipemu::yield(); }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }
// workgroup-scope LDS atomics of the flag protocol in gemm_pc.hip (fibers are cooperative: plain accesses are atomic)
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP      // (the __hip_atomic_* builtins themselves exist in the host compiler too)
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
// counted asynchronous loads (tsii_common.h): queued per thread and retired, oldest first, by async_wait<N> (all but the N newest)
// and by __syncthreads() -- not by lds_barrier(); see emu_runtime.cpp
#define TSII_ASYNC_LOADS 1
typedef float f32x4_emu2 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_emu2 __attribute__((ext_vector_type(4)));
namespace tsii {
static inline void async_load16(f32x4_emu2& d, const void* p) { hipemu::async_issue(&d, p, 16); }
static inline void async_load16(u32x4_emu2& d, const void* p) { hipemu::async_issue(&d, p, 16); }
static inline void async_load4(float& d, const void* p) { hipemu::async_issue(&d, p, 4); }
static inline void async_load16(f32x4_emu2& d, const void* base, unsigned off) { hipemu::async_issue(&d, (const char*)base + off, 16); }
static inline void async_load4(float& d, const void* base, unsigned off) { hipemu::async_issue(&d, (const char*)base + off, 4); }
template <int N, class... T> static inline void async_wait(T&...) { hipemu::async_retire(N); }
static inline void lds_barrier() { hipemu::barrier_only(); }
// direct global -> LDS load: lane i's 16 bytes land at (wave-uniform) base + 16 i when a wait retires them; NaN pattern until then
static inline void async_load16_lds(void* lds_wave_base, const void* gptr) { hipemu::async_issue((char*)lds_wave_base + 16 * hipemu::lane_id(), gptr, 16); }
template <int N> static inline void async_wait_lds() { hipemu::async_retire(N); }
template <int OFF> static inline void lds_read16(u32x4_emu2& d, const void* p) { d = *reinterpret_cast<const u32x4_emu2*>((const char*)p + OFF); }
template <int N, class... T> static inline void lds_wait(T&...) {}
typedef unsigned u32x2_emu2 __attribute__((ext_vector_type(2)));
// ds_read_b64_tr_b16 (bf16_common.h): lane q of a 16-lane group receives element (q & 3) of the 8 bytes at the addresses of lanes 4 i + (q >> 2)
template <int OFF> static inline void lds_read8_tr(u32x2_emu2& d, const void* p) {
    const unsigned long long a = (unsigned long long)((const char*)p + OFF);
    const int l = hipemu::lane_id(), q = l & 15, g = l & ~15;
    unsigned short v[4];
    for (int i = 0; i < 4; ++i) {
        const int src = g + 4 * i + (q >> 2);
        const unsigned lo = __shfl((unsigned)(a & 0xffffffffull), src), hi = __shfl((unsigned)(a >> 32), src);
        v[i] = *reinterpret_cast<const unsigned short*>((((unsigned long long)hi << 32) | lo) + 2 * (q & 3));
    }
    d[0] = v[0] | ((unsigned)v[1] << 16); d[1] = v[2] | ((unsigned)v[3] << 16);
}
template <int N, class... T> static inline void lds_wait4(T&...) {}
}
static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_emu c, int, int, int) {
    return hipemu::mfma_32x32x2(a, b, c);
}
static inline f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int) {
    return hipemu::mfma_16x16x4(a, b, c);
}

static inline f32x16_emu __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf16x8_emu a, bf16x8_emu b, f32x16_emu c, int, int, int) {
    return hipemu::mfma_32x32x16_bf16(a, b, c);
}

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

// ---- runtime API subset -----------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memmove(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
// device properties: a small CU count, so persistent kernels walk several tiles per block in the tests (TSII_EMU_CUS overrides)
#define hipDeviceAttributeMultiprocessorCount 0
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) {
    const char* e = std::getenv("TSII_EMU_CUS");
    *v = e ? std::atoi(e) : 3;
    return 0;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
