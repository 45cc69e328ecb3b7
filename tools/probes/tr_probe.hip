// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS elements does each lane receive?).
// LDS holds bf16-sized 16-bit values equal to their own element index; lane l supplies byte
// address 8*l (natural) in run 0, and a [4 rows x 16 cols] block with row stride 64 elements in run 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(uintptr_t)lds + 8 * l;
    else { int g = l >> 4, i = l & 15; addr = (unsigned)(uintptr_t)lds + 2 * (((g * 4 + (i >> 2)) * 64) + (i & 3) * 4); }
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
