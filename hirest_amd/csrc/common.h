// Shared device helpers for the gfx950 kernels.  CDNA4 only: wave64, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hirest_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define HIREST_WAVE 64

static inline int hirest_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Per-DEVICE one-time kernel configuration (a function attribute set on device 0 says nothing about device 1, and the
// reference's drivers move models with .to(device) without ever calling set_device): raises the dynamic-LDS limit of
// `kern` on the current device the first time it is launched there and reports that device's CU count.
struct HirestDevCfg {
    static constexpr int MAXDEV = 64;
    bool done[MAXDEV] = {};
    int cus[MAXDEV] = {};
};
template <class Kern>
static inline int hirest_configure(Kern kern, int lds_bytes, HirestDevCfg& c, int* cus = nullptr) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= HirestDevCfg::MAXDEV) return HIREST_E_BADARG;
    if (!c.done[dev]) {
        if (lds_bytes > 0) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return (int)e;
        }
        if ((e = hipDeviceGetAttribute(&c.cus[dev], hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
        c.done[dev] = true;
    }
    if (cus) *cus = c.cus[dev];
    return 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// nn.GELU() default (exact erf form, vit_model.py:49) and CLIP's QuickGELU.
//   gelu(x) = x * Phi(x) = max(x, 0) - |x| * Phi(-|x|),   Phi(-a) = 0.5 * erfc(a / sqrt 2) = exp2(Q(a))
// Q = degree-8 polynomial fit of log2(0.5 * erfcx(a / sqrt 2)) on [0, 9.4] with the -a^2/2 * log2(e) term folded into
// its quadratic coefficient.  One v_exp_f32 and 8 FMAs per element (packed two-wide), no division; the older
// Abramowitz-Stegun 7.1.26 form needed v_rcp + v_exp + 10 more ops and cost 12 % of the fc1 GEMM.  Checked against
// x * ndtr(x) in float64 on 2M points of [-12, 12] (tools/gelu_fit.py): max abs error 1.1e-6, relative error <= 2e-5
// everywhere on |x| <= 9.4 (i.e. <= 0.005 bf16 ulp of the value this feeds, tails included).  Beyond 9.4 the
// correction term is clamped (|error| < 3e-20).  This is still the exact-erf GELU, not the tanh approximation.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x), 9.4f);
    float q = 6.909084504513885e-08f;
    q = fmaf(q, a, -3.464947212705738e-06f);
    q = fmaf(q, a, 7.678331166971475e-05f);
    q = fmaf(q, a, -0.0010009667603299022f);
    q = fmaf(q, a, 0.008675649762153625f);
    q = fmaf(q, a, -0.05414620041847229f);
    q = fmaf(q, a, -0.45840057730674744f);
    q = fmaf(q, a, -1.1512391567230225f);
    q = fmaf(q, a, -0.9999977350234985f);
    return fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}
// two-wide form: the Horner chain compiles to v_pk_fma_f32
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), 9.4f), fminf(fabsf(x[1]), 9.4f)};
    f32x2 q = f32x2{6.909084504513885e-08f, 6.909084504513885e-08f} * a + -3.464947212705738e-06f;
    q = q * a + 7.678331166971475e-05f;
    q = q * a + -0.0010009667603299022f;
    q = q * a + 0.008675649762153625f;
    q = q * a + -0.05414620041847229f;
    q = q * a + -0.45840057730674744f;
    q = q * a + -1.1512391567230225f;
    q = q * a + -0.9999977350234985f;
    const f32x2 phi = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 pos = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return pos - a * phi;
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

// The same two reductions on the cross-lane data paths instead of six ds_bpermute round trips (136 vs 680 cycles, same bits:
// tools/probes/wave_sum_probe.hip).  v_permlane32_swap / v_permlane16_swap put a lane's own value and its xor-32 / xor-16 partner's
// into the two registers; the xor-8 / xor-4 steps use DPP row rotations, whose partner lane differs from lane ^ 8 / lane ^ 4 but holds
// the same number (by then a value only depends on the lane index mod 16 / mod 8); xor-2 / xor-1 are DPP quad permutations.  The
// swaps are inline assembly: the builtin, given one value as both operands, reads both results from one register (hipcc 7.2) —
// and `asm volatile`: without it one LayerNorm row in a thousand came out one ulp off in the four-rows-per-wave prologue.
__device__ __forceinline__ void lane_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap32(int& a, int& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap16(int& a, int& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
template <int CTRL> __device__ __forceinline__ float lane_dpp(float v) {      // 0x128 row_ror:8, 0x124 row_ror:4, 0x4E / 0xB1 quad xor 2 / 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int lane_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_sum_x(float v) {
    float a = v, b = v;
    lane_swap32(a, b); v = a + b;
    a = v; b = v;
    lane_swap16(a, b); v = a + b;
    v += lane_dpp<0x128>(v); v += lane_dpp<0x124>(v); v += lane_dpp<0x4E>(v); v += lane_dpp<0xB1>(v);
    return v;
}
__device__ __forceinline__ float wave_max_x(float v) {
    float a = v, b = v;
    lane_swap32(a, b); v = fmaxf(a, b);
    a = v; b = v;
    lane_swap16(a, b); v = fmaxf(a, b);
    v = fmaxf(v, lane_dpp<0x128>(v)); v = fmaxf(v, lane_dpp<0x124>(v)); v = fmaxf(v, lane_dpp<0x4E>(v)); v = fmaxf(v, lane_dpp<0xB1>(v));
    return v;
}

// LayerNorm of one row held by one wave (lane owns float4 number lane + 64 i of the row, zeros past the row's nv = D / 4 vectors):
// two-pass statistics in fp32 like nn.LayerNorm's definition.  Shared by layernorm_rows and the GEMMs that normalise their
// operand rows themselves, so that both give the same bits — which takes the fused multiply-adds written out: left to
// -ffp-contract, `q += d * d` became v_pk_fma_f32 for some unrolled elements and v_pk_mul_f32 + add for others, differently in
// each kernel the function was inlined into (one row in five off by an ulp between the two: tools/f32_stress.py).
template <int NV>
__device__ __forceinline__ void ln_wave_stats(const f32x4 (&v)[NV], int nv, int D, float eps, int lane, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    mean = wave_sum_x(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q = __builtin_fmaf(d, d, q); }
        }
    rstd = 1.0f / sqrtf(wave_sum_x(q) / (float)D + eps);
}
__device__ __forceinline__ f32x4 ln_apply(const f32x4& v, float mean, float rstd, const f32x4& g, const f32x4& b) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((v[e] - mean) * rstd, g[e], b[e]);
    return y;
}

// Async global -> LDS copy of 16 bytes per lane (LDS-DMA).  `lds_wave_base` must be
// wave-uniform; lane i's 16 bytes land at lds_wave_base + 16*i.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
