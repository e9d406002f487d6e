// Probe: lane mapping of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 values equal to their element index; every lane reads 8 bytes
// at a per-lane byte address (mode 0: all lanes address 0; mode 1: lane * 8; mode 2: the [4 rows][16 cols] block pattern with a row
// stride of 64 elements: lane q of a 16-lane group -> row q >> 2, cols 4 (q & 3) ..; groups 1..3 at +16 cols / +4 rows / both).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(int mode, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr = 0;
    if (mode == 1) addr = l * 8;
    if (mode == 2) { const int q = l & 15, g = l >> 4; addr = (((q >> 2) + 4 * (g >> 1)) * 64 + 4 * (q & 3) + 16 * (g & 1)) * 2; }
    addr += (unsigned)(uintptr_t)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
