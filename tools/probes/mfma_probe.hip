// Power-limited MFMA ceiling on MI355X: back-to-back MFMAs on random register operands, no memory traffic.
// Compares v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 (same nominal FLOP rate) at 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ uint32_t hashu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ bf16x8 rnd8(uint32_t seed, int zero) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) {
        uint32_t h = hashu(seed * 8 + i);
        float f = zero ? 0.f : ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
        v[i] = (__bf16)f;
    }
    return v;
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, int zero) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(tid * 16 + i, zero); b[i] = rnd8(tid * 16 + 8 + i, zero); }
    if constexpr (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[tid] = s;
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
        out[tid] = s;
    }
}

template <int SHAPE, int NACC>
void run(const char* name, int threads, int zero) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 2);
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<SHAPE, NACC><<<blocks, threads>>>(out, 2000, zero);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) mfma_loop<SHAPE, NACC><<<blocks, threads>>>(out, iters, zero);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop_per = SHAPE == 32 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2;
    const double flops = (double)blocks * (threads / 64) * iters * NACC * flop_per;
    printf("%-34s waves/CU %d  %s data: %.3f ms  %.0f TFLOP/s\n", name, threads / 64, zero ? "zero  " : "random", ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int zero = 0; zero < 2; ++zero) {
        run<32, 8>("mfma_f32_32x32x16_bf16 x8 acc", 256, zero);
        run<32, 8>("mfma_f32_32x32x16_bf16 x8 acc", 512, zero);
        run<16, 16>("mfma_f32_16x16x32_bf16 x16 acc", 256, zero);
        run<16, 16>("mfma_f32_16x16x32_bf16 x16 acc", 512, zero);
    }
    return 0;
}
