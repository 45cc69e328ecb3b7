// What core clock do small, latency-bound kernels run at?  One wave times a dependent fma chain with both counters: clock64()
// (s_memtime, core clock) and wall_clock64() (constant 100 MHz) -> MHz = 100 * d(clock64) / d(wall_clock64).  Launched once after
// idling, and as a stream of 2000 back-to-back launches (the regime of the step-captioning decode: ~20 small kernels per word).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long* t, int iters) {
    float acc = threadIdx.x, a = 1.000001f, b = 0.5f;
    const long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_fmaf(a, acc, b);
    const long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
    float* o; long* t; hipMalloc(&o, 256); hipMalloc(&t, 16); long h[2];
    auto report = [&](const char* what) { hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); printf("%-52s %6.0f MHz (%ld core cycles in %.1f us)\n", what, 100.0 * h[0] / h[1], h[0], h[1] / 100.0); };
    k<<<1, 64>>>(o, t, 2000); hipDeviceSynchronize(); report("first launch after start-up:");
    for (int i = 0; i < 2000; ++i) k<<<1, 64>>>(o, t, 2000);
    hipDeviceSynchronize(); report("last of 2000 back-to-back one-wave launches:");
    for (int i = 0; i < 2000; ++i) k<<<1024, 256>>>(o, t, 2000);
    hipDeviceSynchronize(); report("last of 2000 back-to-back full-chip launches:");
    k<<<1, 64>>>(o, t, 400000); hipDeviceSynchronize(); report("one wave for a long time:");
    return 0;
}
