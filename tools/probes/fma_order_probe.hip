// What does an fp32 MFMA add, and in which order?  The exact-fp32 kernels (joint.hip) define their results through
// v_mfma_f32_32x32x2_f32 with step j of a 32-deep slab pairing k0 + j (lanes < 32) with k0 + 16 + j (lanes >= 32).  This probe
// computes one 32 x 32 x 192 product (a) with that instruction stream, (b) as a per-element v_fma_f32 chain in the same k order,
// (c) the same chain with the two k of a step swapped, (d) products of a step added first, (e) v_mfma_f32_16x16x4_f32 fed with
// (k0+j, k0+16+j, k0+j+1, k0+16+j+1), (f) v_mfma_f32_4x4x1_f32 one k per instruction — and compares the bits, on benign, wide-
// exponent and denormal-range data.  Build: hipcc --offload-arch=gfx950 -O2 -o fma_order_probe fma_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int K = 192, NS = K / 32;

__global__ void mfma32_kernel(const float* A, const float* W, float* out) {      // out[m][n], 32 x 32
    const int lane = threadIdx.x, row = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int s = 0; s < NS; ++s)
        for (int j = 0; j < 16; ++j) {
            const int k = s * 32 + j + 16 * kh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[row * K + k], A[row * K + k], acc, 0, 0, 0);
        }
    for (int r = 0; r < 16; ++r) out[row * 32 + 8 * (r / 4) + 4 * kh + (r % 4)] = acc[r];
}

template <int MODE>
__global__ void valu_kernel(const float* A, const float* W, float* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 1024) return;
    const int m = t / 32, n = t % 32;
    float acc = 0.f;
    for (int s = 0; s < NS; ++s)
        for (int j = 0; j < 16; ++j) {
            const int k0 = s * 32 + j, k1 = k0 + 16;
            const float a0 = A[m * K + k0], w0 = W[n * K + k0], a1 = A[m * K + k1], w1 = W[n * K + k1];
            if (MODE == 0) { acc = __builtin_fmaf(a0, w0, acc); acc = __builtin_fmaf(a1, w1, acc); }
            else if (MODE == 1) { acc = __builtin_fmaf(a1, w1, acc); acc = __builtin_fmaf(a0, w0, acc); }
            else if (MODE == 2) { acc = acc + __builtin_fmaf(a1, w1, a0 * w0); }
            else { acc = __fadd_rn(acc, __fmul_rn(a0, w0)); acc = __fadd_rn(acc, __fmul_rn(a1, w1)); }   // unfused
        }
    out[m * 32 + n] = acc;
}

__global__ void mfma16_kernel(const float* A, const float* W, float* out) {      // entries m, n < 16
    const int lane = threadIdx.x, idx = lane & 15, ks = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < NS; ++s)
        for (int j = 0; j < 16; j += 2) {
            const int k = s * 32 + j + (ks >> 1) + 16 * (ks & 1);      // slots 0..3 = k0+j, k0+16+j, k0+j+1, k0+16+j+1
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[idx * K + k], A[idx * K + k], acc, 0, 0, 0);
        }
    for (int r = 0; r < 4; ++r) out[idx * 32 + 4 * ks + r] = acc[r];             // D: column (m) = lane % 16, row (n) = 4 (lane / 16) + r
}

__global__ void mfma4_kernel(const float* A, const float* W, float* out) {       // block b: n in 4 (b % 8) .. +3, m in 4 (b / 8) .. +3
    const int lane = threadIdx.x, b = lane >> 2, i = lane & 3;
    const int n = 4 * (b % 8) + i, m = 4 * (b / 8) + i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < NS; ++s)
        for (int j = 0; j < 16; ++j)
            for (int h = 0; h < 2; ++h) {
                const int k = s * 32 + j + 16 * h;
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(W[n * K + k], A[m * K + k], acc, 0, 0, 0);
            }
    for (int r = 0; r < 4; ++r) out[m * 32 + 4 * (b % 8) + r] = acc[r];          // D: column (m) = lane % 4, row (n) = r
}

// latency of a dependent chain: cycles per instruction
__global__ void chain_mfma32(float* out, int iters) {
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = threadIdx.x;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    long t1 = clock64();
    out[threadIdx.x] = acc[0]; if (threadIdx.x == 0) out[64] = float(t1 - t0) / iters;
}
__global__ void chain_mfma16(float* out, int iters) {
    f32x4 acc = {1.f, 2.f, 3.f, 4.f};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    long t1 = clock64();
    out[threadIdx.x] = acc[0]; if (threadIdx.x == 0) out[64] = float(t1 - t0) / iters;
}
__global__ void chain_mfma4(float* out, int iters) {
    f32x4 acc = {1.f, 2.f, 3.f, 4.f};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    long t1 = clock64();
    out[threadIdx.x] = acc[0]; if (threadIdx.x == 0) out[64] = float(t1 - t0) / iters;
}
__global__ void chain_fma(float* out, int iters) {
    float acc = threadIdx.x, a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < iters; ++i) acc = __builtin_fmaf(a, acc, b);
    long t1 = clock64();
    out[threadIdx.x] = acc; if (threadIdx.x == 0) out[64] = float(t1 - t0) / iters;
}

static uint32_t hashu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static void fill(std::vector<float>& v, uint32_t seed, int mode) {
    for (size_t i = 0; i < v.size(); ++i) {
        const uint32_t h = hashu(seed * 1000003u + (uint32_t)i), g = hashu(h ^ 0x9e3779b9u);
        float f = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000);
        if (mode == 1) f = ldexpf(f, (int)(g % 25) - 12);
        if (mode == 2) f = ldexpf(f, -66 - (int)(g % 8));        // products around 2^-132 .. 2^-146: denormal range
        v[i] = f;
    }
}
static void cmp(const char* name, const std::vector<float>& ref, const std::vector<float>& x, int rows, int cols) {
    int bad = 0; long maxulp = 0;
    for (int m = 0; m < rows; ++m)
        for (int n = 0; n < cols; ++n) {
            int32_t a, b; memcpy(&a, &ref[m * 32 + n], 4); memcpy(&b, &x[m * 32 + n], 4);
            if (a != b) { ++bad; long d = labs((long)a - (long)b); if (d > maxulp) maxulp = d; }
        }
    printf("  %-44s %4d of %4d differ (max %ld ulp)\n", name, bad, rows * cols, maxulp);
}

int main() {
    float *dA, *dW, *dO;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dW, 32 * K * 4); hipMalloc(&dO, 1024 * 4);
    const char* names[3] = {"uniform [-1, 1)", "exponents 2^-12 .. 2^12", "denormal-range products"};
    for (int mode = 0; mode < 3; ++mode)
        for (uint32_t seed = 1; seed <= 2; ++seed) {
            std::vector<float> A(32 * K), W(32 * K), ref(1024), x(1024);
            fill(A, seed, mode); fill(W, seed + 77, mode);
            hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), 32 * K * 4, hipMemcpyHostToDevice);
            mfma32_kernel<<<1, 64>>>(dA, dW, dO); hipMemcpy(ref.data(), dO, 4096, hipMemcpyDeviceToHost);
            int nz = 0; for (float f : ref) nz += f != 0.f;
            printf("data: %s, seed %u (reference v_mfma_f32_32x32x2_f32: %d non-zero outputs)\n", names[mode], seed, nz);
            valu_kernel<0><<<4, 256>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("v_fma chain, k0+j then k0+16+j", ref, x, 32, 32);
            valu_kernel<1><<<4, 256>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("v_fma chain, k0+16+j then k0+j", ref, x, 32, 32);
            valu_kernel<2><<<4, 256>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("acc + fma(a1, w1, a0 w0)", ref, x, 32, 32);
            valu_kernel<3><<<4, 256>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("unfused mul + add chain", ref, x, 32, 32);
            hipMemset(dO, 0, 4096); mfma16_kernel<<<1, 64>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("v_mfma_f32_16x16x4_f32 (4 k per instruction)", ref, x, 16, 16);
            hipMemset(dO, 0, 4096); mfma4_kernel<<<1, 64>>>(dA, dW, dO); hipMemcpy(x.data(), dO, 4096, hipMemcpyDeviceToHost); cmp("v_mfma_f32_4x4x1_f32 (1 k per instruction)", ref, x, 8, 32);
        }
    float h[65];
    chain_mfma32<<<1, 64>>>(dO, 4096); hipMemcpy(h, dO, 65 * 4, hipMemcpyDeviceToHost); printf("dependent chain: v_mfma_f32_32x32x2_f32 %.1f cycles per instruction (2 k)\n", h[64]);
    chain_mfma16<<<1, 64>>>(dO, 4096); hipMemcpy(h, dO, 65 * 4, hipMemcpyDeviceToHost); printf("dependent chain: v_mfma_f32_16x16x4_f32 %.1f cycles per instruction (4 k)\n", h[64]);
    chain_mfma4<<<1, 64>>>(dO, 4096); hipMemcpy(h, dO, 65 * 4, hipMemcpyDeviceToHost); printf("dependent chain: v_mfma_f32_4x4x1_f32   %.1f cycles per instruction (1 k)\n", h[64]);
    chain_fma<<<1, 64>>>(dO, 4096); hipMemcpy(h, dO, 65 * 4, hipMemcpyDeviceToHost); printf("dependent chain: v_fma_f32            %.1f cycles per instruction (1 k)\n", h[64]);
    return 0;
}
