// Reference-precision towers: the same EVA ViT / CLIP text transformer forward as tower.hip, but every product in exact fp32
// (v_mfma_f32_32x32x2_f32 GEMMs and flash attention of joint.hip, fp32 LayerNorm, fp32 activations end to end).
//
// Why it exists: the reference runs its encoders in fp32 (EVA_clip/eva_clip.py:90 `precision='fp32'`, modeling.py:120 `.float()`) and
// BASELINE.json asks for retrieval RANKS that match it bit for bit.  The bf16 towers reproduce the embeddings to cos 0.99994, which
// still flips the top-1 of 3 of 546 real prompts whose reference margins (< 3.6e-4) lie below the bf16 score error (1.2e-3).  This
// path is what `precision='fp32'` selects on the host side; it costs ~16x the bf16 path's time (fp32 MFMA peak 157 vs 2500 TFLOP/s)
// and is reported as its own figure, never as the headline.
//
// Workspace (B frames, M = B*T tokens): x f32 [M, D] residual stream, h f32 [M, D] LayerNorm / attention output,
// big f32 [M, max(3D, Dm, kpad)] qkv / MLP hidden / patch rows, eot i32 [B] (text).
#include "common.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct RegionsF { size_t x, h, big, eot, total; };
RegionsF plan_f32(int64_t M, int D, int wide, int B) {
    RegionsF r; size_t off = 0;
    r.x = off; off += align256((size_t)M * D * 4);
    r.h = off; off += align256((size_t)M * D * 4);
    r.big = off; off += align256((size_t)M * wide * 4);
    r.eot = off; off += align256((size_t)B * 4);
    r.total = off;
    return r;
}

inline int wide_of(int D, int Dm, int kpad) { int w = 3 * D; if (Dm > w) w = Dm; if (kpad > w) w = kpad; return w; }

#define CHECK(expr) do { int _e = (expr); if (_e != 0) return _e; } while (0)

// im2col with one all-zero row in front of every frame's patches: row b*T + 1 + p = patch p of frame b (fp32), row b*T = zeros
// (the CLS position; overwritten after the GEMM).  Column k = c*P*P + ph*P + pw, columns >= 3*P*P zero (vit_model.py:198,205).
template <int IN_DTYPE>
__global__ __launch_bounds__(256) void patch_rows_f32_kernel(const void* __restrict__ frames, int B, int S, int P,
                                                            const float* __restrict__ mean3, const float* __restrict__ std3,
                                                            float* __restrict__ rows, int Kpad) {
    const int G = S / P, PP = P * P, K = 3 * PP, T = G * G + 1;
    const int chunks = Kpad >> 2;
    const int64_t total = (int64_t)B * T * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const int64_t row = idx / chunks;
        const int t = (int)(row % T), b = (int)(row / T);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            const int pw_i = (t - 1) % G, ph_i = (t - 1) / G;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = ch * 4 + e;
                if (k < K) {
                    const int c = k / PP, rr = k - c * PP;
                    const int dy = rr / P, dx = rr - dy * P;
                    const int y = ph_i * P + dy, xx = pw_i * P + dx;
                    if constexpr (IN_DTYPE == 0) {
                        o[e] = reinterpret_cast<const float*>(frames)[(((int64_t)b * 3 + c) * S + y) * S + xx];
                    } else if constexpr (IN_DTYPE == 1) {
                        o[e] = (float)reinterpret_cast<const bf16_t*>(frames)[(((int64_t)b * 3 + c) * S + y) * S + xx];
                    } else {
                        const float u = (float)reinterpret_cast<const uint8_t*>(frames)[(((int64_t)b * S + y) * S + xx) * 3 + c];
                        o[e] = (u / 255.0f - mean3[c]) / std3[c];
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(rows + row * Kpad + 4 * ch) = o;
    }
}

// one pre-LN block in fp32: x += proj(attn(LN1 x)); x += fc2(act(fc1(LN2 x)))   (vit_model.py:175-182, eva_model.py:110-159)
int run_block_f32(const hirest_block_weights_f32& w, float* x, float* h, float* big, int B, int T, int D, int heads, int dh, int Dm,
                  float eps, int act, int causal, void* stream) {
    const int M = B * T;
    CHECK(hirest_layernorm(x, D, nullptr, w.ln1_g, w.ln1_b, eps, h, D, 1, M, D, stream));
    CHECK(hirest_gemm_f32(h, D, w.qkv_w, D, w.qkv_b, nullptr, 0, nullptr, 0, big, 3 * D, M, 3 * D, D, 0, stream));
    // the text tower's additive causal mask is -inf above the diagonal (eva_model.py:224-230): a penalty whose exp is exactly 0
    CHECK(hirest_attention_f32_qkv(big, 3 * (int64_t)D, big + D, big + 2 * D, 3 * (int64_t)D, h, B, T, T, heads, dh,
                                   1.0f / sqrtf((float)dh), 0.f, causal ? -1.0e30f : 0.f, stream));
    CHECK(hirest_gemm_f32(h, D, w.proj_w, D, w.proj_b, x, D, nullptr, 0, x, D, M, D, D, 0, stream));
    CHECK(hirest_layernorm(x, D, nullptr, w.ln2_g, w.ln2_b, eps, h, D, 1, M, D, stream));
    CHECK(hirest_gemm_f32(h, D, w.fc1_w, D, w.fc1_b, nullptr, 0, nullptr, 0, big, Dm, M, Dm, D, act == 1 ? 3 : 1, stream));
    CHECK(hirest_gemm_f32(big, Dm, w.fc2_w, Dm, w.fc2_b, x, D, nullptr, 0, x, D, M, D, Dm, 0, stream));
    return 0;
}

}  // namespace

extern "C" size_t hirest_vision_workspace_bytes_f32(const hirest_vision_tower_f32* t, int32_t B) {
    if (!t || B <= 0) return 0;
    const int T = (t->image_size / t->patch) * (t->image_size / t->patch) + 1;
    return plan_f32((int64_t)B * T, t->width, wide_of(t->width, t->mlp_dim, t->kpad), B).total;
}

// Front end of the fp32 (and bf16x3) vision towers: x[b*T + 0] = cls + pos[0], x[b*T + 1 + p] = conv(patch p) + bias + pos[1 + p], then
// ln_pre when the tower has one (vit_model.py:198-205,330-333; model.py:229-238).  `rows` is scratch of [B*T, kpad] floats.
extern "C" int hirest_vision_embed_f32(const hirest_vision_tower_f32* t, const void* frames, int32_t in_dtype, int32_t B, float* x,
                                       float* rows, void* stream) {
    if (!t || !frames || !x || !rows || B <= 0) return HIREST_E_BADARG;
    if (t->image_size % t->patch != 0 || t->kpad % 16 != 0 || t->kpad < 3 * t->patch * t->patch) return HIREST_E_SHAPE;
    if (in_dtype < 0 || in_dtype > 2 || (in_dtype == 2 && (!t->image_mean || !t->image_std))) return HIREST_E_BADARG;
    const int G = t->image_size / t->patch, T = G * G + 1, D = t->width;
    const int64_t M = (int64_t)B * T;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // patch embedding: rows b*T + 1 + p = conv(patch p) + bias + pos[1 + p]; the zero row b*T is then overwritten by cls + pos[0]
    {
        const int64_t total = M * (t->kpad / 4);
        int64_t blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
        dim3 grid((unsigned)blocks), block(256);
        switch (in_dtype) {
            case 0: hipLaunchKernelGGL(patch_rows_f32_kernel<0>, grid, block, 0, s, frames, B, t->image_size, t->patch, t->image_mean, t->image_std, rows, t->kpad); break;
            case 1: hipLaunchKernelGGL(patch_rows_f32_kernel<1>, grid, block, 0, s, frames, B, t->image_size, t->patch, t->image_mean, t->image_std, rows, t->kpad); break;
            default: hipLaunchKernelGGL(patch_rows_f32_kernel<2>, grid, block, 0, s, frames, B, t->image_size, t->patch, t->image_mean, t->image_std, rows, t->kpad); break;
        }
        if (int e = hirest_launch_status()) return e;
    }
    CHECK(hirest_gemm_f32(rows, t->kpad, t->patch_w, t->kpad, t->patch_b, nullptr, 0, t->pos, T, x, D, (int)M, D, t->kpad, 0, stream));
    CHECK(hirest_write_cls_rows(x, D, t->cls, t->pos, B, T, D, stream));
    if (t->ln_pre_g)
        CHECK(hirest_layernorm(x, D, nullptr, t->ln_pre_g, t->ln_pre_b, t->ln_eps, x, D, 1, (int)M, D, stream));
    return 0;
}

extern "C" int hirest_vision_forward_f32(const hirest_vision_tower_f32* t, const void* frames, int32_t in_dtype, int32_t B, float* out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!t || !frames || !out || !workspace || B <= 0 || !t->blocks) return HIREST_E_BADARG;
    if (t->width != t->heads * t->head_dim || t->image_size % t->patch != 0 || t->kpad % 16 != 0 || t->kpad < 3 * t->patch * t->patch)
        return HIREST_E_SHAPE;
    if (in_dtype < 0 || in_dtype > 2 || (in_dtype == 2 && (!t->image_mean || !t->image_std))) return HIREST_E_BADARG;
    const int G = t->image_size / t->patch, T = G * G + 1, D = t->width;
    const int64_t M = (int64_t)B * T;
    const RegionsF r = plan_f32(M, D, wide_of(D, t->mlp_dim, t->kpad), B);
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + r.x);
    float* h = reinterpret_cast<float*>(ws + r.h);
    float* big = reinterpret_cast<float*>(ws + r.big);
    CHECK(hirest_vision_embed_f32(t, frames, in_dtype, B, x, big, stream));
    for (int l = 0; l < t->layers; ++l)
        CHECK(run_block_f32(t->blocks[l], x, h, big, B, T, D, t->heads, t->head_dim, t->mlp_dim, t->ln_eps, t->act, 0, stream));
    if (t->out_all_tokens) {
        CHECK(hirest_layernorm(x, D, nullptr, t->norm_g, t->norm_b, t->ln_eps, h, D, 1, (int)M, D, stream));
        CHECK(hirest_gemm_f32(h, D, t->head_w, D, t->head_b, nullptr, 0, nullptr, 0, out, t->embed_dim, (int)M, t->embed_dim, D, 0, stream));
        return 0;
    }
    CHECK(hirest_layernorm(x, (int64_t)T * D, nullptr, t->norm_g, t->norm_b, t->ln_eps, h, D, 1, B, D, stream));
    CHECK(hirest_gemm_f32(h, D, t->head_w, D, t->head_b, nullptr, 0, nullptr, 0, out, t->embed_dim, B, t->embed_dim, D, 0, stream));
    return 0;
}

extern "C" size_t hirest_text_workspace_bytes_f32(const hirest_text_tower_f32* t, int32_t B) {
    if (!t || B <= 0) return 0;
    return plan_f32((int64_t)B * t->context, t->width, 4 * t->width, B).total;
}

extern "C" int hirest_text_forward_f32(const hirest_text_tower_f32* t, const int64_t* tokens, int32_t B, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!t || !tokens || !out || !workspace || B <= 0 || !t->blocks) return HIREST_E_BADARG;
    if (t->width % t->heads != 0) return HIREST_E_SHAPE;
    const int L = t->context, D = t->width, dh = D / t->heads;
    const RegionsF r = plan_f32((int64_t)B * L, D, 4 * D, B);
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + r.x);
    float* h = reinterpret_cast<float*>(ws + r.h);
    float* big = reinterpret_cast<float*>(ws + r.big);
    int32_t* eot = reinterpret_cast<int32_t*>(ws + r.eot);
    CHECK(hirest_embed_tokens(tokens, t->tok_emb, t->pos, x, eot, B, L, D, t->vocab, stream));
    for (int l = 0; l < t->layers; ++l)
        CHECK(run_block_f32(t->blocks[l], x, h, big, B, L, D, t->heads, dh, 4 * D, t->ln_eps, t->act, 1, stream));
    CHECK(hirest_layernorm(x, D, eot, t->lnf_g, t->lnf_b, t->ln_eps, h, D, 1, B, D, stream));
    CHECK(hirest_gemm_f32(h, D, t->proj_w, D, nullptr, nullptr, 0, nullptr, 0, out, t->embed_dim, B, t->embed_dim, D, 0, stream));
    return 0;
}
