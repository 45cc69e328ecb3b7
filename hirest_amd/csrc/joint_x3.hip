// Joint model, split-operand precision ("bf16x3", include/hirest_hip.h: hirest_joint_encoder_x3): the clip4caption VisualModel encoder
// (module_visual.py:104-264, 396-424; called by modeling.py:196-211 once per moment-retrieval batch and twenty times per segmentation
// batch) with its six linear layers per block on the bf16 matrix pipe — three MFMAs per product on bf16 hi + lo splits of both fp32
// operands — instead of v_mfma_f32_*_f32 at 1/16 of the bf16 rate.  What is NOT a weight product stays the fp32 path's kernel: the flash
// attention with the uniform -10000 shift (joint.hip), LayerNorm, the residual adds, the heads.
//
// A post-LN block needs its LayerNorm output twice: as fp32 (the next residual) and as the split A operand of the next GEMM; the LayerNorm
// kernel below writes both in one pass.  The GELU epilogue of intermediate.dense writes the split format directly (the 3072-wide hidden
// activation never exists in fp32); the attention output is split by one extra pass.  All GEMMs run on the 128 x 128 split-operand kernel
// (HIREST_GEMM_X3_T128): at 1 500 rows (B = 5, T = 300) a 768- / 2304- / 3072-wide layer is 72 / 216 / 288 tiles for 256 CUs.
#include "common.h"
#include "profile.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
#define CHECK(expr) do { int _e = (expr); if (_e != 0) return _e; } while (0)

__device__ __forceinline__ void split_store4(const f32x4& y, bf16_t* orow, int col) {   // col % 4 == 0: 4 columns of one 32-block
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)y[e]; lo[e] = (bf16_t)(y[e] - (float)hi[e]); }
    bf16_t* o = orow + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + 32) = lo;
}

// one wave per row, four rows per block (layernorm_rows' arithmetic: ln_wave_stats / ln_apply)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_f32_split2_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ add, int period,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                  float* __restrict__ out32, int64_t ldo32, bf16_t* __restrict__ out2, int64_t ldo2,
                                                                  int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    f32x4 g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i, cc = c < nv ? c : nv - 1;
        g[i] = *reinterpret_cast<const f32x4*>(gamma + 4 * cc);
        b[i] = *reinterpret_cast<const f32x4*>(beta + 4 * cc);
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float* xr = x + (int64_t)row * ldx;
        const float* ar = add ? add + (int64_t)(row % period) * D : nullptr;
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nv ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (ar && c < nv) v[i] += *reinterpret_cast<const f32x4*>(ar + 4 * c);
        }
        float mean, rstd;
        ln_wave_stats<NV>(v, nv, D, eps, lane, mean, rstd);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 y = ln_apply(v[i], mean, rstd, g[i], b[i]);
                if (out32) *reinterpret_cast<f32x4*>(out32 + (int64_t)row * ldo32 + 4 * c) = y;
                if (out2) split_store4(y, out2 + (int64_t)row * ldo2, 4 * c);
            }
        }
    }
}

int gemm_x3(const hirest_bf16* A2, int64_t lda, const hirest_bf16* W2, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
            int K, int epi, void* stream, void* splitk_scratch = nullptr) {
    hirest_gemm_args a;
    a.struct_size = sizeof(a);
    a.A = A2; a.lda = lda; a.W = W2; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo;
    a.M = M; a.N = N; a.K = 2 * K; a.epilogue = epi; a.pos = nullptr; a.patches_per_frame = 0; a.aux0 = splitk_scratch; a.aux1 = nullptr;
    a.flags = HIREST_GEMM_X3 | HIREST_GEMM_X3_T128;
    return hirest_gemm_bf16(&a, stream);
}

struct RegionsJ { size_t f2, x, a, x2, big, ctx, h2, part, total; };
RegionsJ plan(int64_t M, int D, int Dm, int Din) {
    RegionsJ r; size_t off = 0;
    r.f2 = off; off += align256((size_t)M * 2 * Din * 2);
    r.x = off; off += align256((size_t)M * D * 4);
    r.a = off; off += align256((size_t)M * D * 4);
    r.x2 = off; off += align256((size_t)M * 2 * D * 2);
    r.big = off; off += align256((size_t)M * 3 * D * 4);
    r.ctx = off; off += align256((size_t)M * D * 4);
    r.h2 = off; off += align256((size_t)M * 2 * Dm * 2);
    r.part = off; off += align256((size_t)4 * M * D * 4);           // split-K partial tiles of the two 768-wide residual GEMMs (up to 4 slices)
    r.total = off;
    return r;
}

bool shape_ok(const hirest_joint_encoder_x3* e) {
    return e->layers > 0 && e->heads > 0 && e->width > 0 && e->width % e->heads == 0 && e->width % 32 == 0 && e->mlp_dim % 32 == 0 &&
           e->in_dim % 32 == 0 && e->width <= 2048;
}

}  // namespace

extern "C" int hirest_layernorm_f32_split2(const float* x, int64_t ldx, const float* add, int32_t period, const float* gamma, const float* beta,
                                           float eps, float* out32, int64_t ldo32, hirest_bf16* out2, int64_t ldo2, int32_t rows, int32_t D,
                                           void* stream) {
    if (!x || !gamma || !beta || (!out32 && !out2) || rows <= 0 || (add && period <= 0)) return HIREST_E_BADARG;
    if (D <= 0 || D % 32 != 0 || D > 2048 || ldx % 4 != 0) return HIREST_E_SHAPE;
    if (out32 && ldo32 % 4 != 0) return HIREST_E_SHAPE;
    if (out2 && (ldo2 % 8 != 0 || ldo2 < 2 * (int64_t)D)) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bf16_t* o2 = reinterpret_cast<bf16_t*>(out2);
    const int nv = (D / 4 + 63) / 64;
    int grid = (rows + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
    HirestProfScope pr(HIREST_PROF_LAYERNORM, 13, rows, D, 0, s);
#define LS_CASE(NVV) case NVV: hipLaunchKernelGGL((layernorm_f32_split2_kernel<NVV>), dim3(grid), dim3(256), 0, s, x, ldx, add, period, gamma, beta, eps, \
                                                  out32, ldo32, o2, ldo2, rows, D); break;
    switch (nv) { LS_CASE(1) LS_CASE(2) LS_CASE(3) LS_CASE(4) LS_CASE(5) LS_CASE(6) LS_CASE(7) LS_CASE(8) default: return HIREST_E_SHAPE; }
#undef LS_CASE
    return hirest_launch_status();
}

extern "C" size_t hirest_joint_encoder_x3_workspace_bytes(const hirest_joint_encoder_x3* e, int32_t B, int32_t T) {
    if (!e || e->struct_size != sizeof(*e) || B <= 0 || T <= 0 || !shape_ok(e)) return 0;
    return plan((int64_t)B * T, e->width, e->mlp_dim, e->in_dim).total;
}

extern "C" int hirest_joint_encoder_x3_forward(const hirest_joint_encoder_x3* e, const float* f, int32_t B, int32_t T, float* out,
                                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || e->struct_size != sizeof(*e) || !f || !out || !workspace || B <= 0 || T <= 0 || !e->layer || !e->emb_w2 || !e->pos) return HIREST_E_BADARG;
    if (!shape_ok(e) || T > e->max_pos) return HIREST_E_SHAPE;
    const int64_t M64 = (int64_t)B * T;
    if (M64 > 0x7fffffff) return HIREST_E_SHAPE;
    const int M = (int)M64, D = e->width, Dm = e->mlp_dim, Din = e->in_dim, dh = D / e->heads;
    const RegionsJ r = plan(M64, D, Dm, Din);
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    hirest_bf16* f2 = reinterpret_cast<hirest_bf16*>(ws + r.f2);
    float* x = reinterpret_cast<float*>(ws + r.x);
    float* a = reinterpret_cast<float*>(ws + r.a);
    hirest_bf16* x2 = reinterpret_cast<hirest_bf16*>(ws + r.x2);
    float* big = reinterpret_cast<float*>(ws + r.big);
    float* ctx = reinterpret_cast<float*>(ws + r.ctx);
    hirest_bf16* h2 = reinterpret_cast<hirest_bf16*>(ws + r.h2);
    void* part = ws + r.part;
    const float scale = 1.0f / sqrtf((float)dh);
    // embeddings: word_embeddings (a Linear) + position rows, LayerNorm (module_visual.py:56-81)
    CHECK(hirest_split2_bf16(f, Din, f2, 2 * Din, M, Din, 0, stream));
    CHECK(gemm_x3(f2, 2 * Din, e->emb_w2, 2 * Din, e->emb_b, big, D, M, D, Din, HIREST_EPI_BIAS_F32, stream));
    float* cur = e->layers % 2 == 0 ? out : x;                    // the block outputs alternate between two buffers; the last one lands in `out`
    float* other = cur == out ? x : out;
    CHECK(hirest_layernorm_f32_split2(big, D, e->pos, T, e->emb_ln_g, e->emb_ln_b, e->ln_eps, cur, D, x2, 2 * D, M, D, stream));
    for (int l = 0; l < e->layers; ++l) {
        const hirest_joint_layer_x3& w = e->layer[l];
        const bool last = l + 1 == e->layers;
        CHECK(gemm_x3(x2, 2 * D, w.qkv_w2, 2 * D, w.qkv_b, big, 3 * D, M, 3 * D, D, HIREST_EPI_BIAS_F32, stream));
        CHECK(hirest_attention_f32(big, ctx, B, T, e->heads, dh, scale, e->attn_shift, stream));
        CHECK(hirest_split2_bf16(ctx, D, x2, 2 * D, M, D, 0, stream));
        // cur += ctx Wo^T + b (the residual of attention.output), then LayerNorm -> a (fp32) and x2 (split)
        CHECK(gemm_x3(x2, 2 * D, w.ao_w2, 2 * D, w.ao_b, cur, D, M, D, D, HIREST_EPI_BIAS_RESID_F32, stream, part));
        CHECK(hirest_layernorm_f32_split2(cur, D, nullptr, 0, w.ln1_g, w.ln1_b, e->ln_eps, a, D, x2, 2 * D, M, D, stream));
        CHECK(gemm_x3(x2, 2 * D, w.fc1_w2, 2 * D, w.fc1_b, h2, 2 * Dm, M, Dm, D, HIREST_EPI_BIAS_GELU_SPLIT2, stream));
        CHECK(gemm_x3(h2, 2 * Dm, w.fc2_w2, 2 * Dm, w.fc2_b, a, D, M, D, Dm, HIREST_EPI_BIAS_RESID_F32, stream, part));
        CHECK(hirest_layernorm_f32_split2(a, D, nullptr, 0, w.ln2_g, w.ln2_b, e->ln_eps, other, D, last ? nullptr : x2, 2 * D, M, D, stream));
        float* t = cur; cur = other; other = t;
    }
    return cur == out ? 0 : HIREST_E_BADARG;                      // (by construction)
}
