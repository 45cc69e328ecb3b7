// Probe: semantics of v_permlane16_swap_b32 on gfx950.  a = 1000 + lane, b = 2000 + lane; prints, per 16-lane row,
// which source each result holds.  Expected (ISA text): odd rows of the first operand are exchanged with even rows of the
// second: a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3].
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(1000u + l, 2000u + l, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 128 * 4);
    unsigned h[128];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w)
        for (int row = 0; row < 4; ++row)
            printf("%s' row %d: lanes hold %u .. %u\n", w ? "b" : "a", row, h[w * 64 + row * 16], h[w * 64 + row * 16 + 15]);
    return 0;
}
