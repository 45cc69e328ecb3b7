// wave_sum by ds_bpermute shuffles (v += shfl_xor(v, 32), 16, 8, 4, 2, 1) against the same tree on the cross-lane data paths
// (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8 / row_ror:4 / quad_perm): the partner lanes differ for the 8 / 4 steps, but
// by then a lane's value only depends on its index mod 16 / mod 8, so every lane adds the same two numbers -> same bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ float slow_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ float slow_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
template <int CTRL> __device__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// (the swaps are written as inline assembly: with the same value as both operands the builtin's two results are taken from one
// register by hipcc 7.2 and nothing moves)
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ float fast_sum(float v) {
    float a = v, b = v;
    swap32(a, b); v = a + b;
    a = v; b = v;
    swap16(a, b); v = a + b;
    v += dpp<0x128>(v); v += dpp<0x124>(v); v += dpp<0x4E>(v); v += dpp<0xB1>(v);
    return v;
}
__device__ float fast_max(float v) {
    float a = v, b = v;
    swap32(a, b); v = fmaxf(a, b);
    a = v; b = v;
    swap16(a, b); v = fmaxf(a, b);
    v = fmaxf(v, dpp<0x128>(v)); v = fmaxf(v, dpp<0x124>(v)); v = fmaxf(v, dpp<0x4E>(v)); v = fmaxf(v, dpp<0xB1>(v));
    return v;
}
__global__ void k(const float* in, float* out, long* cyc) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    float v = in[t];
    long t0 = clock64(); float a = v; for (int i = 0; i < 16; ++i) a = slow_sum(a) * 0.015625f; long t1 = clock64();
    float b = v; for (int i = 0; i < 16; ++i) b = fast_sum(b) * 0.015625f; long t2 = clock64();
    if (a != b) cyc[2] = 1;
    a = slow_sum(v); b = fast_sum(v);
    out[4 * t] = a; out[4 * t + 1] = b; out[4 * t + 2] = slow_max(v); out[4 * t + 3] = fast_max(v);
    if (t == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    const int n = 64 * 4096;
    float* h = new float[n]; uint32_t s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; int e = (int)((s >> 8) % 40) - 20; s = s * 1664525u + 1013904223u;
        h[i] = ldexpf(((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)), e); }
    float *din, *dout; long* dc; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 16); hipMalloc(&dc, 32); hipMemset(dc, 0, 32);
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 64, 64>>>(din, dout, dc);
    float* o = new float[4 * n]; long c[2];
    hipMemcpy(o, dout, n * 16, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    int bad_sum = 0, bad_max = 0;
    for (int i = 0; i < n; ++i) { bad_sum += memcmp(&o[4 * i], &o[4 * i + 1], 4) != 0; bad_max += memcmp(&o[4 * i + 2], &o[4 * i + 3], 4) != 0; }
    printf("wave_sum: %d of %d lanes differ; wave_max: %d differ; cycles per wave_sum: shuffle %ld vs cross-lane %ld\n", bad_sum, n, bad_max, c[0] / 16, c[1] / 16);
    return 0;
}
