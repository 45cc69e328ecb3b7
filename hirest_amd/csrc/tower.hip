// Tower runners: enqueue one full forward of a pre-LN transformer tower on a stream.
//   vision: EVA ViT  (vit_model.py:326-351)   frames -> [B, embed_dim]
//   text:   CLIP text transformer (eva_model.py:232-250)   token ids -> [B, embed_dim]
// Pure orchestration of the kernels in gemm/attention/elementwise; no allocation, no sync.
//
// Workspace (B frames, M = B*T tokens), every region 256-B aligned:
//   x    f32  [M, D]                     residual stream (fp32: 40 residual adds stay exact-ish)
//   h    bf16 [M, D]                     LN output / attention output (never live together)
//   big  bf16 [M, max(3D, Dm, kpad*)]    qkv / MLP hidden / patches (never live together)
//   eot  i32  [B]                        text only
//   xb / part / stats                     vision calls of >= 64 frames with folded-LayerNorm weights: see plan()
#include "common.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Regions {
    size_t x, h, big, eot, xb, xl, cls, part, stats, guard, total;
};

// lnfold: also the folded-LayerNorm buffers (vision): xb bf16 [M,D] = rounded copy of the residual stream (A operand of
// qkv / fc1) and, since round 4, its HIGH part: between the first and the last block the stream lives as xb + xl (bf16 hi + bf16 lo,
// 16 significand bits; HIREST_EPI_BIAS_RESID2_LNSTATS) instead of in x — 8 instead of 10 bytes of epilogue traffic per element on the
// two byte-bound residual GEMMs of a block; HIREST_TOWER_F32_RESIDUAL keeps the fp32 array; part f32 [M, ceil(D/64), 2] row sums per 64-column group, stats f32 [M,2] (mean, rstd)
Regions plan(int64_t M, int D, int wide, int B, bool lnfold = false) {
    Regions r;
    size_t off = 0;
    r.x = off; off += align256((size_t)M * D * 4);
    r.h = off; off += align256((size_t)M * D * 2);
    r.big = off; off += align256((size_t)M * wide * 2);
    r.eot = off; off += align256((size_t)B * 4);
    r.xb = r.xl = r.cls = r.part = r.stats = r.guard = off;
    if (lnfold) {
        r.xb = off; off += align256((size_t)M * D * 2);
        r.xl = off; off += align256((size_t)M * D * 2);   // low part of the two-array residual stream (hi = xb)
        r.cls = off; off += 2 * align256((size_t)B * D * 2);   // the CLS rows' hi | lo, compact, for the pruned last block
        r.part = off; off += align256((size_t)M * ((D + 63) / 64) * 8);
        r.stats = off; off += align256((size_t)M * 8);
        r.guard = off; off += 256;                      // one float: max |mean| / sigma over all rows and layers of the call
    }
    r.total = off;
    return r;
}

// The folded path needs the persistent GEMM kernels (large problems) and the row-statistics kernel's width limit; below
// 64 frames the attention also switches kernels, so 64 is the one boundary where results may differ within rounding.
inline bool vision_lnfold(const hirest_vision_tower* t, int B) {
    if (!t->blocks || !t->blocks[0].qkv_wf || t->act != 0 || t->ln_pre_g || B < 64) return false;
    const int T = (t->image_size / t->patch) * (t->image_size / t->patch) + 1;
    const int64_t M = (int64_t)B * T;
    return t->width >= 256 && t->width <= 1536 && t->width % 8 == 0 && M * t->width >= (int64_t)2048 * 1024;
}

inline int vision_wide(const hirest_vision_tower* t) {
    int w = 3 * t->width;
    if (t->mlp_dim > w) w = t->mlp_dim;
    if (t->kpad > w) w = t->kpad;
    return w;
}

#define CHECK(expr) do { int _e = (expr); if (_e != 0) return _e; } while (0)

int gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
         int K, int epi, void* stream, const float* pos = nullptr, int P = 0, void* aux0 = nullptr, void* aux1 = nullptr,
         int flags = 0) {
    hirest_gemm_args a;
    a.struct_size = sizeof(a);
    a.A = reinterpret_cast<const hirest_bf16*>(A); a.lda = lda;
    a.W = reinterpret_cast<const hirest_bf16*>(W); a.ldw = ldw;
    a.bias = bias; a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.epilogue = epi;
    a.pos = pos; a.patches_per_frame = P; a.aux0 = aux0; a.aux1 = aux1; a.flags = flags;
    return hirest_gemm_bf16(&a, stream);
}

// one pre-LN block: x += proj(attn(LN1 x)); x += fc2(act(fc1(LN2 x)))
int run_block(const hirest_block_weights& w, float* x, hirest_bf16* h, hirest_bf16* big, int B, int T, int D, int heads,
              int dh, int Dm, float eps, int act, int causal, void* stream) {
    const int M = B * T;
    CHECK(hirest_layernorm(x, D, nullptr, w.ln1_g, w.ln1_b, eps, h, D, 0, M, D, stream));
    CHECK(gemm(h, D, w.qkv_w, D, w.qkv_b, big, 3 * D, M, 3 * D, D, HIREST_EPI_BIAS_BF16, stream));
    CHECK(hirest_attention_bf16(big, h, B, T, heads, dh, 1.0f / sqrtf((float)dh), causal, stream));
    CHECK(gemm(h, D, w.proj_w, D, w.proj_b, x, D, M, D, D, HIREST_EPI_BIAS_RESID_F32, stream));
    CHECK(hirest_layernorm(x, D, nullptr, w.ln2_g, w.ln2_b, eps, h, D, 0, M, D, stream));
    CHECK(gemm(h, D, w.fc1_w, D, w.fc1_b, big, Dm, M, Dm, D,
               act == 1 ? HIREST_EPI_BIAS_QGELU_BF16 : HIREST_EPI_BIAS_GELU_BF16, stream));
    CHECK(gemm(big, Dm, w.fc2_w, Dm, w.fc2_b, x, D, M, D, Dm, HIREST_EPI_BIAS_RESID_F32, stream));
    return 0;
}

// the same block with both LayerNorms folded into the GEMMs around them (include/hirest_hip.h, HIREST_EPI_LNFOLD_*):
// on entry xb / stats describe x; proj and fc2 refresh them in their epilogues
// xl != nullptr: the residual stream is xb + xl (see plan()); x is not touched
int run_block_lnfold(const hirest_block_weights& w, float* x, hirest_bf16* h, hirest_bf16* big, hirest_bf16* xb, float* part,
                     float* stats, float* guard, int B, int T, int D, int heads, int dh, int Dm, float eps, void* stream,
                     hirest_bf16* xl = nullptr) {
    const int M = B * T, G = (D + 63) / 64;
    void* res = xl ? static_cast<void*>(xl) : static_cast<void*>(x);
    const int epi_res = xl ? HIREST_EPI_BIAS_RESID2_LNSTATS : HIREST_EPI_BIAS_RESID_LNSTATS_F32;
    CHECK(gemm(xb, D, w.qkv_wf, D, w.qkv_bf, big, 3 * D, M, 3 * D, D, HIREST_EPI_LNFOLD_BF16, stream, nullptr, 0, stats,
               const_cast<float*>(w.qkv_s)));
    CHECK(hirest_attention_bf16(big, h, B, T, heads, dh, 1.0f / sqrtf((float)dh), 0, stream));
    CHECK(gemm(h, D, w.proj_w, D, w.proj_b, res, D, M, D, D, epi_res, stream, nullptr, 0, xb, part));
    CHECK(hirest_ln_stats_finalize(part, G, stats, eps, M, D, guard, stream));
    CHECK(gemm(xb, D, w.fc1_wf, D, w.fc1_bf, big, Dm, M, Dm, D, HIREST_EPI_LNFOLD_GELU_BF16, stream, nullptr, 0, stats,
               const_cast<float*>(w.fc1_s)));
    // fc2 walks its rows backwards: it starts on the part of the hidden activation fc1 wrote last, and the next block's qkv
    // (forwards) starts on the rows of xb fc2 wrote last — both still in the 256-MB Infinity Cache.  (Alternating every
    // kernel of the chain, attention included, measured less: the grouped tile order of qkv / fc1 runs slower backwards.)
    CHECK(gemm(big, Dm, w.fc2_w, Dm, w.fc2_b, res, D, M, D, Dm, epi_res, stream, nullptr, 0, xb, part, HIREST_GEMM_REVERSE));
    CHECK(hirest_ln_stats_finalize(part, G, stats, eps, M, D, guard, stream));
    return 0;
}

// The LAST block of a tower whose caller reads only token 0 of every frame (vit_model.py:340-351: norm -> x[:, 0] -> head):
// keys and values are still needed for all tokens, but the attention output, proj, fc1 and fc2 only for the B CLS rows.  The
// qkv GEMM runs as usual; attention computes the leading query tile of each frame; the three GEMMs after it take the CLS rows
// through strides (A / residual rows T*D apart) with M = B, and the row-indexed side buffers (xb, part, stats) are used
// compactly, row b = frame b — nothing reads their all-token contents after the qkv GEMM.  Every surviving row goes through
// the same kernels and epilogues as in run_block_lnfold, so the CLS rows of x are bit-identical to the unpruned block's.
int run_block_lnfold_cls(const hirest_block_weights& w, float* x, hirest_bf16* h, hirest_bf16* big, hirest_bf16* xb, float* part,
                         float* stats, float* guard, int B, int T, int D, int heads, int dh, int Dm, float eps, void* stream,
                         hirest_bf16* xl = nullptr, hirest_bf16* cls_hi = nullptr, hirest_bf16* cls_lo = nullptr) {
    const int M = B * T, G = (D + 63) / 64;
    const int64_t TD = (int64_t)T * D;
    CHECK(gemm(xb, D, w.qkv_wf, D, w.qkv_bf, big, 3 * D, M, 3 * D, D, HIREST_EPI_LNFOLD_BF16, stream, nullptr, 0, stats,
               const_cast<float*>(w.qkv_s)));
    CHECK(hirest_attention_bf16_rows(big, h, B, T, heads, dh, 1.0f / sqrtf((float)dh), 0, 1, stream));
    if (xl) {
        // two-array residual stream: the CLS rows' hi / lo move into compact [B, D] arrays (the epilogue indexes its hi array by output
        // row), and the block updates those — the same arithmetic per row as run_block_lnfold's, so the result equals the unpruned block's
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        if (hipMemcpy2DAsync(cls_hi, (size_t)D * 2, xb, (size_t)TD * 2, (size_t)D * 2, B, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(cls_lo, (size_t)D * 2, xl, (size_t)TD * 2, (size_t)D * 2, B, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return hirest_launch_status();
        CHECK(gemm(h, TD, w.proj_w, D, w.proj_b, cls_lo, D, B, D, D, HIREST_EPI_BIAS_RESID2_LNSTATS, stream, nullptr, 0, cls_hi, part));
        CHECK(hirest_ln_stats_finalize(part, G, stats, eps, B, D, guard, stream));
        CHECK(gemm(cls_hi, D, w.fc1_wf, D, w.fc1_bf, big, Dm, B, Dm, D, HIREST_EPI_LNFOLD_GELU_BF16, stream, nullptr, 0, stats,
                   const_cast<float*>(w.fc1_s)));
        CHECK(gemm(big, Dm, w.fc2_w, Dm, w.fc2_b, cls_lo, D, B, D, Dm, HIREST_EPI_BIAS_RESID2_LNSTATS, stream, nullptr, 0, cls_hi, part));
        return 0;
    }
    CHECK(gemm(h, TD, w.proj_w, D, w.proj_b, x, TD, B, D, D, HIREST_EPI_BIAS_RESID_LNSTATS_F32, stream, nullptr, 0, xb, part));
    CHECK(hirest_ln_stats_finalize(part, G, stats, eps, B, D, guard, stream));
    CHECK(gemm(xb, D, w.fc1_wf, D, w.fc1_bf, big, Dm, B, Dm, D, HIREST_EPI_LNFOLD_GELU_BF16, stream, nullptr, 0, stats,
               const_cast<float*>(w.fc1_s)));
    CHECK(gemm(big, Dm, w.fc2_w, Dm, w.fc2_b, x, TD, B, D, Dm, HIREST_EPI_BIAS_RESID_LNSTATS_F32, stream, nullptr, 0, xb, part));
    return 0;
}

}  // namespace

extern "C" int hirest_abi_version(void) { return HIREST_ABI_VERSION; }

extern "C" const char* hirest_build_info(void) {
    return "hirest_hip gfx950 (CDNA4) | gemm pq256 (persistent ping-pong, 2 phases per K step, 16x16x32 bf16 MFMA) / p256 / t128 + LDS-DMA | "
           "bf16x3 split-operand GEMM + attention | attention 16x16x32 bf16 MFMA + tr16 reads | " __VERSION__;
}

extern "C" size_t hirest_vision_workspace_bytes(const hirest_vision_tower* t, int32_t B) {
    if (!t || B <= 0) return 0;
    const int T = (t->image_size / t->patch) * (t->image_size / t->patch) + 1;
    return plan((int64_t)B * T, t->width, vision_wide(t), B, vision_lnfold(t, B)).total;
}

extern "C" size_t hirest_vision_guard_offset(const hirest_vision_tower* t, int32_t B) {
    if (!t || B <= 0 || !vision_lnfold(t, B)) return (size_t)-1;
    const int T = (t->image_size / t->patch) * (t->image_size / t->patch) + 1;
    return plan((int64_t)B * T, t->width, vision_wide(t), B, true).guard;
}

extern "C" int hirest_vision_forward(const hirest_vision_tower* t, const void* frames, int32_t in_dtype, int32_t B,
                                     float* out, void* workspace, size_t workspace_bytes, int32_t flags, void* stream) {
    if (!t || !frames || !out || !workspace || B <= 0 || !t->blocks) return HIREST_E_BADARG;
    if (t->width != t->heads * t->head_dim || t->image_size % t->patch != 0) return HIREST_E_SHAPE;
    const int G = t->image_size / t->patch, P = G * G, T = P + 1, D = t->width;
    const bool can_fold = vision_lnfold(t, B), lnfold = can_fold && !(flags & HIREST_TOWER_NO_LNFOLD);
    const Regions r = plan((int64_t)B * T, D, vision_wide(t), B, can_fold);   // one layout for both forms of a call
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + r.x);
    hirest_bf16* h = reinterpret_cast<hirest_bf16*>(ws + r.h);
    hirest_bf16* big = reinterpret_cast<hirest_bf16*>(ws + r.big);

    // patch-embed: im2col -> GEMM whose epilogue adds bias + pos and scatters to rows b*T+1+p; CLS rows
    CHECK(hirest_patchify(frames, in_dtype, B, t->image_size, t->patch, t->image_mean, t->image_std, big, t->kpad, stream));
    CHECK(gemm(big, t->kpad, t->patch_w, t->kpad, t->patch_b, x, D, B * P, D, t->kpad, HIREST_EPI_PATCH_POS_F32, stream,
               t->pos, P));
    CHECK(hirest_write_cls_rows(x, D, t->cls, t->pos, B, T, D, stream));
    if (t->ln_pre_g)   // model.py:261 ln_pre, in place on the fp32 stream (row-local, so in place is safe)
        CHECK(hirest_layernorm(x, D, nullptr, t->ln_pre_g, t->ln_pre_b, t->ln_eps, x, D, 1, B * T, D, stream));
    int64_t cls_ldx = (int64_t)T * D;                         // row stride of the CLS rows the head reads
    if (lnfold) {
        hirest_bf16* xb = reinterpret_cast<hirest_bf16*>(ws + r.xb);
        float* part = reinterpret_cast<float*>(ws + r.part);
        float* stats = reinterpret_cast<float*>(ws + r.stats);
        float* guard = reinterpret_cast<float*>(ws + r.guard);
        if (hipMemsetAsync(guard, 0, 256, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return hirest_launch_status();
        const bool prune = !t->out_all_tokens && !(flags & HIREST_TOWER_NO_PRUNE);
        // the two-array residual stream between the first and the last block (a one-block tower with pruning never leaves x)
        hirest_bf16* xl = ((flags & HIREST_TOWER_F32_RESIDUAL) || (prune && t->layers == 1)) ? nullptr : reinterpret_cast<hirest_bf16*>(ws + r.xl);
        hirest_bf16* cls_hi = reinterpret_cast<hirest_bf16*>(ws + r.cls);
        hirest_bf16* cls_lo = reinterpret_cast<hirest_bf16*>(ws + r.cls + align256((size_t)B * D * 2));
        CHECK(hirest_rowstats_split_bf16(x, D, xb, xl, stats, t->ln_eps, B * T, D, guard, stream));
        for (int l = 0; l < t->layers; ++l) {
            if (prune && l == t->layers - 1) {
                CHECK(run_block_lnfold_cls(t->blocks[l], x, h, big, xb, part, stats, guard, B, T, D, t->heads, t->head_dim, t->mlp_dim,
                                           t->ln_eps, stream, xl, cls_hi, cls_lo));
            } else {
                CHECK(run_block_lnfold(t->blocks[l], x, h, big, xb, part, stats, guard, B, T, D, t->heads, t->head_dim, t->mlp_dim,
                                       t->ln_eps, stream, xl));
            }
        }
        if (xl && !prune) CHECK(hirest_combine_hi_lo_f32(xb, xl, D, x, D, B * T, D, stream));
        if (xl && prune) {                                     // the CLS rows back to fp32, compact [B, D] at the start of x
            CHECK(hirest_combine_hi_lo_f32(cls_hi, cls_lo, D, x, D, B, D, stream));
            cls_ldx = D;
        }
    } else {
        for (int l = 0; l < t->layers; ++l)
            CHECK(run_block(t->blocks[l], x, h, big, B, T, D, t->heads, t->head_dim, t->mlp_dim, t->ln_eps, t->act, 0, stream));
    }
    if (t->out_all_tokens) {   // ln_post + proj on every token row; out is [B*T, embed_dim]
        CHECK(hirest_layernorm(x, D, nullptr, t->norm_g, t->norm_b, t->ln_eps, h, D, 0, B * T, D, stream));
        CHECK(gemm(h, D, t->head_w, D, t->head_b, out, t->embed_dim, B * T, t->embed_dim, D, HIREST_EPI_BIAS_F32, stream));
        return 0;
    }
    // norm on the CLS rows only (LayerNorm is per-row, so norm(x)[:,0] == norm(x[:,0])), then head
    CHECK(hirest_layernorm(x, cls_ldx, nullptr, t->norm_g, t->norm_b, t->ln_eps, h, D, 0, B, D, stream));
    CHECK(gemm(h, D, t->head_w, D, t->head_b, out, t->embed_dim, B, t->embed_dim, D, HIREST_EPI_BIAS_F32, stream));
    return 0;
}

extern "C" size_t hirest_text_workspace_bytes(const hirest_text_tower* t, int32_t B) {
    if (!t || B <= 0) return 0;
    return plan((int64_t)B * t->context, t->width, 4 * t->width, B).total;
}

extern "C" int hirest_text_forward(const hirest_text_tower* t, const int64_t* tokens, int32_t B, float* out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!t || !tokens || !out || !workspace || B <= 0 || !t->blocks) return HIREST_E_BADARG;
    if (t->width % t->heads != 0) return HIREST_E_SHAPE;
    const int L = t->context, D = t->width, dh = D / t->heads;
    const Regions r = plan((int64_t)B * L, D, 4 * D, B);
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + r.x);
    hirest_bf16* h = reinterpret_cast<hirest_bf16*>(ws + r.h);
    hirest_bf16* big = reinterpret_cast<hirest_bf16*>(ws + r.big);
    int32_t* eot = reinterpret_cast<int32_t*>(ws + r.eot);

    CHECK(hirest_embed_tokens(tokens, t->tok_emb, t->pos, x, eot, B, L, D, t->vocab, stream));
    for (int l = 0; l < t->layers; ++l)
        CHECK(run_block(t->blocks[l], x, h, big, B, L, D, t->heads, dh, 4 * D, t->ln_eps, t->act, 1, stream));
    // ln_final on the EOT rows only, then @ text_projection
    CHECK(hirest_layernorm(x, D, eot, t->lnf_g, t->lnf_b, t->ln_eps, h, D, 0, B, D, stream));
    CHECK(gemm(h, D, t->proj_w, D, nullptr, out, t->embed_dim, B, t->embed_dim, D, HIREST_EPI_BIAS_F32, stream));
    return 0;
}
