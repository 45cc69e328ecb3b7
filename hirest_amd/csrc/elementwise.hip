// HBM-bound row kernels around the GEMMs: LayerNorm, patch extraction, CLS/pos prologue,
// token embedding + EOT index, dtype casts.  One wave per row, 16-B vector accesses.
#include "common.h"
#include "profile.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (float4 per lane per step), two-pass
// statistics in fp32 (mean, then E[(x-mean)^2]) exactly like nn.LayerNorm's definition.
// ---------------------------------------------------------------------------------------------
template <int NV, bool OUT_F32>   // NV = ceil(D/4/64) float4 per lane
__global__ __launch_bounds__(256) void layernorm_rows(const float* __restrict__ x, int64_t ldx,
                                                     const int32_t* __restrict__ row_index,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, void* __restrict__ out, int64_t ldo, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t src = row_index ? (int64_t)row_index[row] : (int64_t)row;
    const float* xr = x + src * ldx;
    const int nv = D >> 2;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nv ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float mean, rstd;
    ln_wave_stats<NV>(v, nv, D, eps, lane, mean, rstd);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 y = ln_apply(v[i], mean, rstd, *reinterpret_cast<const f32x4*>(gamma + 4 * c), *reinterpret_cast<const f32x4*>(beta + 4 * c));
            if constexpr (OUT_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (int64_t)row * ldo + 4 * c) = y;
            } else {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16_t)y[e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + (int64_t)row * ldo + 4 * c) = o;
            }
        }
    }
}

// Bulk variant for the per-layer LayerNorms (rows >> CUs): each wave owns R consecutive rows and issues
// all of their loads before the first reduction, so R x NV x 16 B per lane are in flight (the one-row
// kernel is latency-bound at ~2.9 TB/s); grid-strided so the launch is a few thousand blocks.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_rows_bulk(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, bf16_t* __restrict__ out, int64_t ldo, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    f32x4 g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        g[i] = c < nv ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        b[i] = c < nv ? *reinterpret_cast<const f32x4*>(beta + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row0 = wave_global * R; row0 < rows; row0 += nwaves * R) {
        f32x4 v[R][NV];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < rows ? row0 + r : rows - 1;
            const float* xr = x + (int64_t)row * ldx;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                v[r][i] = c < nv ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
            const float mean = wave_sum(s) / (float)D;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (lane + 64 * i < nv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mean; q += d * d; }
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
            if (row0 + r < rows) {
                bf16_t* orow = out + (int64_t)(row0 + r) * ldo;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nv) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((v[r][i][e] - mean) * rstd * g[i][e] + b[i][e]);
                        __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(orow + 4 * c));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs (HIREST_EPI_LNFOLD_*): the statistics side.
//   rowstats_bf16_kernel   x f32 [M,D] -> xb = bf16(x) and (mean, rstd) of the ROUNDED row (start of a tower call; inside
//                          the layer loop the residual GEMMs' epilogues produce both)
//   ln_finalize_kernel     per-row (sum, sum of squares) of G 32-column groups -> (mean, rstd); 8 lanes per row, the G
//                          partials are combined in double so the E[x^2] - mean^2 form loses nothing to their order
// ---------------------------------------------------------------------------------------------
// Guard of the folded form: max over rows of |mean| * rstd = |mean| / sigma of the rounded row.  The fold feeds the GEMM
// the UN-normalised bf16 row, so a row-wide offset of r sigma costs r / 2^9 sigma of rounding error per element where the
// LayerNorm pass would have cost 2^-9 of the normalised value; the host falls back to the LayerNorm pass for a call whose
// worst row exceeds its threshold (hirest_vision_forward, HIREST_TOWER_NO_LNFOLD).  Non-negative floats order like their
// bit patterns, so the maximum is an integer atomicMax; the plain read in front keeps the steady state atomic-free.
__device__ __forceinline__ void guard_max(float* guard, float v) {
    if (v > __builtin_nontemporal_load(guard)) atomicMax(reinterpret_cast<int*>(guard), __float_as_int(v));
}

template <int NV>
__global__ __launch_bounds__(256) void rowstats_bf16_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ xb,
                                                            float* __restrict__ stats, float eps, int rows, int D,
                                                            float* __restrict__ guard, bf16_t* __restrict__ xlo) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    float worst = 0.f;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float* xr = x + (int64_t)row * ldx;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * c));
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)v[e]; const float f = (float)o[e]; s += f; q = fmaf(f, f, q); }
                *reinterpret_cast<bf16x4*>(xb + (int64_t)row * D + 4 * c) = o;
                if (xlo) {                                            // the two-array residual stream: lo = bf16(x - hi)
                    bf16x4 l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) l[e] = (bf16_t)(v[e] - (float)o[e]);
                    *reinterpret_cast<bf16x4*>(xlo + (int64_t)row * D + 4 * c) = l;
                }
            }
        }
        const double S = (double)wave_sum(s), Q = (double)wave_sum(q);
        const double mean = S / D;
        double var = Q / D - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * (int64_t)row) = f32x2{(float)mean, rstd};
        const float r = fabsf((float)mean) * rstd;
        worst = r <= 3.0e38f ? fmaxf(worst, r) : __builtin_inff();   // a NaN / inf row trips the guard (fmaxf alone would drop a NaN)
    }
    if (guard && lane == 0) guard_max(guard, worst);
}

__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int G, float* __restrict__ stats,
                                                          float eps, int rows, int D, float* __restrict__ guard) {
    const int sub = threadIdx.x & 7;
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    double S = 0.0, Q = 0.0;
    if (row < rows) {
        const f32x2* pr = reinterpret_cast<const f32x2*>(part) + row * G;
        for (int g = sub; g < G; g += 8) { const f32x2 v = pr[g]; S += (double)v[0]; Q += (double)v[1]; }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { S += __shfl_xor(S, m); Q += __shfl_xor(Q, m); }
    float ratio = 0.f;
    if (row < rows && sub == 0) {
        const double mean = S / D;
        double var = Q / D - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        *reinterpret_cast<f32x2*>(stats + 2 * row) = f32x2{(float)mean, rstd};
        ratio = fabsf((float)mean) * rstd;
        ratio = ratio <= 3.0e38f ? ratio : __builtin_inff();          // NaN / inf rows trip the guard
    }
    if (guard) {
        ratio = wave_max(ratio);
        if ((threadIdx.x & 63) == 0) guard_max(guard, ratio);
    }
}

// LayerNorm folded into the Linear that follows it, weight preparation (once per checkpoint; HIREST_EPI_LNFOLD_*):
//   Wf[n][k] = bf16(W[n][k] * gamma[k]),  s[n] = sum_k float(Wf[n][k]),  b'[n] = b[n] + sum_k W[n][k] * beta[k]
// One wave per output row n; sums in double (they are one-time and feed every row of every call).
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ Wf, float* __restrict__ bias_out, float* __restrict__ s_out,
                                                             int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* wr = W + (int64_t)n * K;
    double ss = 0.0, sb = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float w = wr[k];
        const bf16_t wf = (bf16_t)(w * gamma[k]);
        Wf[(int64_t)n * K + k] = wf;
        ss += (double)(float)wf;
        sb += (double)w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); sb += __shfl_xor(sb, o, 64); }
    if (lane == 0) { s_out[n] = (float)ss; bias_out[n] = (float)((double)(bias ? bias[n] : 0.f) + sb); }
}

// ---------------------------------------------------------------------------------------------
// Patch extraction.  One thread produces 8 consecutive k-columns (16 B of bf16) of one patch row.
// Column k = c*P*P + ph*P + pw (conv weight flatten order, vit_model.py:198).
// ---------------------------------------------------------------------------------------------
template <int IN_DTYPE>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ frames, int B, int S, int P,
                                                      const float* __restrict__ mean3, const float* __restrict__ std3,
                                                      bf16_t* __restrict__ patches, int Kpad) {
    const int G = S / P, PP = P * P, K = 3 * PP;
    const int chunks = Kpad >> 3;
    const int64_t total = (int64_t)B * G * G * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const int64_t prow = idx / chunks;
        const int pw_i = (int)(prow % G);
        const int ph_i = (int)((prow / G) % G);
        const int b = (int)(prow / ((int64_t)G * G));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            float val = 0.f;
            if (k < K) {
                const int c = k / PP, rr = k - c * PP;
                const int dy = rr / P, dx = rr - dy * P;
                const int y = ph_i * P + dy, xx = pw_i * P + dx;
                if constexpr (IN_DTYPE == 0) {
                    val = reinterpret_cast<const float*>(frames)[(((int64_t)b * 3 + c) * S + y) * S + xx];
                } else if constexpr (IN_DTYPE == 1) {
                    val = (float)reinterpret_cast<const bf16_t*>(frames)[(((int64_t)b * 3 + c) * S + y) * S + xx];
                } else {
                    const float u = (float)reinterpret_cast<const uint8_t*>(frames)[(((int64_t)b * S + y) * S + xx) * 3 + c];
                    val = (u / 255.0f - mean3[c]) / std3[c];
                }
            }
            o[e] = (bf16_t)val;
        }
        *reinterpret_cast<bf16x8*>(patches + prow * Kpad + ch * 8) = o;
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ x, int64_t ldx, const float* __restrict__ cls,
                                                      const float* __restrict__ pos0, int B, int T, int D) {
    const int nv = D >> 2;
    const int64_t total = (int64_t)B * nv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv);
        const int b = (int)(idx / nv);
        const f32x4 a = *reinterpret_cast<const f32x4*>(cls + 4 * c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos0 + 4 * c);
        *reinterpret_cast<f32x4*>(x + (int64_t)b * T * ldx + 4 * c) = a + p;
    }
}

// one wave per (b,t) row; wave 0 of each batch row's first block also computes the EOT index
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ tok_emb,
                                                          const float* __restrict__ pos, float* __restrict__ x,
                                                          int32_t* __restrict__ eot_row, int B, int L, int D, int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, t = row - b * L;
    int64_t id = tokens[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* e = tok_emb + id * D;
    const float* p = pos + (int64_t)t * D;
    float* o = x + (int64_t)row * D;
    for (int c = lane; c < (D >> 2); c += 64)
        *reinterpret_cast<f32x4*>(o + 4 * c) = *reinterpret_cast<const f32x4*>(e + 4 * c) + *reinterpret_cast<const f32x4*>(p + 4 * c);
    if (t == 0 && eot_row) {
        // argmax over the L token ids, first maximum (torch.argmax semantics on distinct maxima)
        int64_t best = -1; int bi = 0;
        for (int i = lane; i < L; i += 64) {
            const int64_t v = tokens[(int64_t)b * L + i];
            if (v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            const int64_t ov = __shfl_xor(best, o2, 64);
            const int oi = __shfl_xor(bi, o2, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) eot_row[b] = b * L + bi;
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + 4 * i);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x4*>(out + 4 * i) = o;
    }
}

inline int grid_for(int64_t total, int block = 256, int cap = 256 * 8) {
    int64_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <bool OUT_F32>
int launch_ln(const float* x, int64_t ldx, const int32_t* ri, const float* g, const float* b, float eps, void* out,
              int64_t ldo, int rows, int D, hipStream_t s) {
    int nv = (D / 4 + 63) / 64;
    if (nv > 8) nv = nv <= 12 ? 12 : (nv <= 16 ? 16 : (nv <= 24 ? 24 : 32));  // round up to an instantiated size
    dim3 grid((rows + 3) / 4), block(256);
#define LN_CASE(NVV) \
    case NVV: hipLaunchKernelGGL((layernorm_rows<NVV, OUT_F32>), grid, block, 0, s, x, ldx, ri, g, b, eps, out, ldo, rows, D); break;
    switch (nv) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        LN_CASE(12) LN_CASE(16) LN_CASE(24) LN_CASE(32)
        default: return HIREST_E_SHAPE;
    }
#undef LN_CASE
    return hirest_launch_status();
}

}  // namespace

extern "C" int hirest_layernorm(const float* x, int64_t ldx, const int32_t* row_index, const float* gamma,
                                const float* beta, float eps, void* out, int64_t ldo, int32_t out_is_f32,
                                int32_t rows, int32_t D, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0) return HIREST_E_BADARG;
    if (D <= 0 || D % 4 != 0 || D > 8192 || ldx % 4 != 0 || ldo % 4 != 0) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HirestProfScope prof(HIREST_PROF_LAYERNORM, 0, rows, D, 0, s);
    if (!out_is_f32 && !row_index && rows >= 8192 && D <= 6 * 256) {   // the per-layer LayerNorms
        const int nv = (D / 4 + 63) / 64;
        const int grid = 256 * 8;
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
#define LNB_CASE(NVV) \
    case NVV: hipLaunchKernelGGL((layernorm_rows_bulk<NVV, 4>), dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, eps, o, ldo, rows, D); \
        return hirest_launch_status();
        switch (nv) { LNB_CASE(1) LNB_CASE(2) LNB_CASE(3) LNB_CASE(4) LNB_CASE(5) LNB_CASE(6) default: break; }
#undef LNB_CASE
    }
    if (out_is_f32) return launch_ln<true>(x, ldx, row_index, gamma, beta, eps, out, ldo, rows, D, s);
    return launch_ln<false>(x, ldx, row_index, gamma, beta, eps, out, ldo, rows, D, s);
}

// x = hi + lo in fp32 for the rows row_stride apart (the two-array residual stream back to fp32: all rows after a tower's last block, or its
// CLS rows only); one thread per 4 columns
__global__ __launch_bounds__(256) void combine_hi_lo_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int64_t ld_in,
                                                            float* __restrict__ out, int64_t ldo, int rows, int D) {
    const int nv = D >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * nv; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nv;
        const int c = (int)(i - r * nv) * 4;
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(hi + r * ld_in + c), b = *reinterpret_cast<const bf16x4*>(lo + r * ld_in + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)a[e] + (float)b[e];
        *reinterpret_cast<f32x4*>(out + r * ldo + c) = v;
    }
}

extern "C" int hirest_combine_hi_lo_f32(const hirest_bf16* hi, const hirest_bf16* lo, int64_t ld_in, float* out, int64_t ldo, int32_t rows,
                                        int32_t D, void* stream) {
    if (!hi || !lo || !out || rows <= 0 || D <= 0) return HIREST_E_BADARG;
    if (D % 4 != 0 || ld_in % 4 != 0 || ldo % 4 != 0) return HIREST_E_SHAPE;
    int64_t blocks = ((int64_t)rows * (D / 4) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(combine_hi_lo_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(hi), reinterpret_cast<const bf16_t*>(lo), ld_in, out, ldo, rows, D);
    return hirest_launch_status();
}

extern "C" int hirest_rowstats_bf16(const float* x, int64_t ldx, hirest_bf16* xb, float* stats, float eps, int32_t rows, int32_t D,
                                    float* guard, void* stream) {
    return hirest_rowstats_split_bf16(x, ldx, xb, nullptr, stats, eps, rows, D, guard, stream);
}

extern "C" int hirest_rowstats_split_bf16(const float* x, int64_t ldx, hirest_bf16* xb, hirest_bf16* xlo, float* stats, float eps, int32_t rows,
                                          int32_t D, float* guard, void* stream) {
    if (!x || !xb || !stats || rows <= 0) return HIREST_E_BADARG;
    if (D <= 0 || D % 4 != 0 || D > 6 * 256 || ldx % 4 != 0) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bf16_t* o = reinterpret_cast<bf16_t*>(xb);
    const int nv = (D / 4 + 63) / 64;
    const int grid = rows / 4 + 1 < 256 * 16 ? rows / 4 + 1 : 256 * 16;
#define RS_CASE(NVV) case NVV: hipLaunchKernelGGL((rowstats_bf16_kernel<NVV>), dim3(grid), dim3(256), 0, s, x, ldx, o, stats, eps, rows, D, guard, reinterpret_cast<bf16_t*>(xlo)); break;
    switch (nv) { RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) default: return HIREST_E_SHAPE; }
#undef RS_CASE
    return hirest_launch_status();
}

extern "C" int hirest_ln_stats_finalize(const float* partials, int32_t groups, float* stats, float eps, int32_t rows, int32_t D,
                                        float* guard, void* stream) {
    if (!partials || !stats || rows <= 0 || groups <= 0 || D <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partials, groups,
                       stats, eps, rows, D, guard);
    return hirest_launch_status();
}

extern "C" int hirest_patchify(const void* frames, int32_t in_dtype, int32_t B, int32_t S, int32_t P,
                               const float* mean3, const float* std3, hirest_bf16* patches, int32_t Kpad, void* stream) {
    if (!frames || !patches || B <= 0 || S <= 0 || P <= 0) return HIREST_E_BADARG;
    if (S % P != 0 || Kpad % 8 != 0 || Kpad < 3 * P * P) return HIREST_E_SHAPE;
    if (in_dtype == 2 && (!mean3 || !std3)) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int G = S / P;
    const int64_t total = (int64_t)B * G * G * (Kpad / 8);
    dim3 grid(grid_for(total, 256, 256 * 16)), block(256);
    bf16_t* o = reinterpret_cast<bf16_t*>(patches);
    switch (in_dtype) {
        case 0: hipLaunchKernelGGL(patchify_kernel<0>, grid, block, 0, s, frames, B, S, P, mean3, std3, o, Kpad); break;
        case 1: hipLaunchKernelGGL(patchify_kernel<1>, grid, block, 0, s, frames, B, S, P, mean3, std3, o, Kpad); break;
        case 2: hipLaunchKernelGGL(patchify_kernel<2>, grid, block, 0, s, frames, B, S, P, mean3, std3, o, Kpad); break;
        default: return HIREST_E_BADARG;
    }
    return hirest_launch_status();
}

extern "C" int hirest_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, hirest_bf16* Wf,
                                     float* bias_out, float* colsum_out, int32_t N, int32_t K, void* stream) {
    if (!W || !gamma || !beta || !Wf || !bias_out || !colsum_out || N <= 0 || K <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(fold_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, gamma, beta, bias,
                       reinterpret_cast<bf16_t*>(Wf), bias_out, colsum_out, N, K);
    return hirest_launch_status();
}

extern "C" int hirest_write_cls_rows(float* x, int64_t ldx, const float* cls, const float* pos0, int32_t B,
                                     int32_t tokens_per_frame, int32_t D, void* stream) {
    if (!x || !cls || !pos0 || B <= 0) return HIREST_E_BADARG;
    if (D % 4 != 0 || ldx % 4 != 0) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for((int64_t)B * (D / 4))), dim3(256), 0, s, x, ldx, cls, pos0, B,
                       tokens_per_frame, D);
    return hirest_launch_status();
}

extern "C" int hirest_embed_tokens(const int64_t* tokens, const float* tok_emb, const float* pos, float* x,
                                   int32_t* eot_row, int32_t B, int32_t L, int32_t D, int32_t vocab, void* stream) {
    if (!tokens || !tok_emb || !pos || !x || B <= 0 || L <= 0) return HIREST_E_BADARG;
    if (D % 4 != 0) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((B * L + 3) / 4), dim3(256), 0, s, tokens, tok_emb, pos, x, eot_row, B, L, D, vocab);
    return hirest_launch_status();
}

extern "C" int hirest_f32_to_bf16(const float* in, hirest_bf16* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) return HIREST_E_BADARG;
    if (n % 4 != 0) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, in, reinterpret_cast<bf16_t*>(out), n / 4);
    return hirest_launch_status();
}
