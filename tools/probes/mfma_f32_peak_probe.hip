// The fp32 matrix pipe's sustained rate, chip-wide: every SIMD of every CU issues v_mfma_f32_32x32x2_f32 (or 16x16x4) from registers
// only (4 independent accumulators per wave, 1 / 2 / 4 waves per SIMD), with random or zero operands, for ~1 ms — the ceiling the
// 157-TF paper figure (256 flop / clk / CU at 2.4 GHz) turns into on this part, and the core clock it runs at (clock64 vs the 100-MHz
// wall clock).  The fp32 GEMM kernels of csrc/joint.hip are priced against what this prints.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, long* t, int iters, float scale) {
    const float a = scale * (1.0f + 0.001f * (threadIdx.x % 61)), b = scale * (0.5f + 0.002f * (threadIdx.x % 53));
    const long c0 = clock64(), w0 = wall_clock64();
    float r = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) r += acc[u][e];
    } else {
        f32x4 acc[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
        }
        for (int u = 0; u < 4; ++u) for (int e = 0; e < 4; ++e) r += acc[u][e];
    }
    const long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int SHAPE>
void run(const char* what, int blocks, int iters, float scale, float* o, long* t) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<blocks, 256>>>(o, t, iters, scale); hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k<SHAPE><<<blocks, 256>>>(o, t, iters, scale);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double flop = (SHAPE == 32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2) * 4 * iters * 4.0 * blocks * reps;   // 4 MFMAs x 4 waves per block
    printf("%-34s %5d blocks  %7.1f TFLOP/s   core clock %5.0f MHz   (%.2f ms per launch)\n", what, blocks, flop / (ms * 1e-3) / 1e12,
           100.0 * h[0] / h[1], ms / reps);
}
int main() {
    float* o; long* t; hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&t, 16);
    for (int bl : {256, 512, 1024}) {
        run<32>("32x32x2 f32, random operands", bl, 4000 * 1024 / bl, 1.0f, o, t);
        run<32>("32x32x2 f32, zero operands", bl, 4000 * 1024 / bl, 0.0f, o, t);
        run<16>("16x16x4 f32, random operands", bl, 16000 * 1024 / bl, 1.0f, o, t);
    }
    return 0;
}
