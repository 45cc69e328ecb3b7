// Step-captioning decoder, one beam-search step per call (clip4caption DecoderModel: module_decoder.py:279-406, driven by
// train.py:511-599).  Pure orchestration, like tower.hip: the call enqueues ~35 kernels of joint.hip / elementwise.hip / train.hip on
// the stream — embedding of every beam's newest token, two post-LN decoder layers (self-attention over the beam's kept K / V,
// cross-attention to the 20 encoded frames, FFN), the LM head and log-softmax + running beam score — without going back to the host
// in between (the host loop used to spend more time issuing these launches one by one than the GPU spent running them).
//
// K / V cache: the reference re-runs the whole prefix every step (train.py:547-566).  Its "causal" penalty is -10000 added to the
// scores of future keys (module_decoder.py:394-397); exp() of that is exactly 0 in fp32, so a position's hidden state never depends
// on later tokens and the rows kept here are bit for bit the rows a recomputation would produce (every kernel involved is
// batch-invariant).  Beams are re-ordered every step, so each row's history is gathered from its PARENT row of the previous step.
#include "common.h"

namespace {

// dst[r][j] = j < t ? src[parent[r]][j] : new[r]   for K and V of one layer; rows of D floats; src holds t positions per beam,
// dst t + 1 (both compact: the whole history is rewritten every step because the beams are re-ordered every step)
__global__ void kv_gather_append_kernel(const float* __restrict__ k_src, const float* __restrict__ v_src,
                                        const int32_t* __restrict__ parent, const float* __restrict__ qkv_new, float* __restrict__ k_dst,
                                        float* __restrict__ v_dst, int R, int t, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one float4 each
    const int d4 = D >> 2;
    if (i >= (int64_t)R * (t + 1) * d4) return;
    const int c = (int)(i % d4);
    const int64_t rj = i / d4;
    const int j = (int)(rj % (t + 1)), r = (int)(rj / (t + 1));
    f32x4 kv, vv;
    if (j < t) {
        const int64_t s = ((int64_t)parent[r] * t + j) * D + 4 * c;
        kv = *reinterpret_cast<const f32x4*>(k_src + s);
        vv = *reinterpret_cast<const f32x4*>(v_src + s);
    } else {
        kv = *reinterpret_cast<const f32x4*>(qkv_new + (int64_t)r * 3 * D + D + 4 * c);
        vv = *reinterpret_cast<const f32x4*>(qkv_new + (int64_t)r * 3 * D + 2 * D + 4 * c);
    }
    const int64_t o = ((int64_t)r * (t + 1) + j) * D + 4 * c;
    *reinterpret_cast<f32x4*>(k_dst + o) = kv;
    *reinterpret_cast<f32x4*>(v_dst + o) = vv;
}

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Beam bookkeeping of one step for every sample (clip4caption/modules/beam.py:70-92 given the device's top-`beam` of the
// beam x vocabulary scores, which arrive sorted, so the reference's torch.sort of the scores is the identity).  One block per
// sample, thread k = beam k.  A finished sample (top beam emitted [SEP] in an earlier step) is left untouched and gets inert rows.
__global__ void beam_advance_kernel(const float* __restrict__ val, const int32_t* __restrict__ idx, int beam, int vocab, int step,
                                    int max_steps, int eos, float* __restrict__ scores, int32_t* __restrict__ tokens,
                                    int32_t* __restrict__ backptr, int32_t* __restrict__ n_steps, int32_t* __restrict__ done,
                                    int32_t* __restrict__ next_ids, int32_t* __restrict__ next_parents, float* __restrict__ next_add) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= beam) return;
    const int row = b * beam + k;
    if (done[b]) {                                       // (block-uniform: written only by this block, in an earlier launch)
        next_ids[row] = eos; next_parents[row] = row; next_add[row] = 0.f;
        return;
    }
    const int flat = idx[row];
    const int prev = flat / vocab, word = flat - prev * vocab;
    scores[row] = val[row];
    tokens[((int64_t)b * max_steps + step) * beam + k] = word;
    backptr[((int64_t)b * max_steps + step) * beam + k] = prev;
    next_ids[row] = word; next_parents[row] = b * beam + prev; next_add[row] = val[row];
    if (k == 0) {
        n_steps[b] = step + 1;
        if (word == eos) done[b] = 1;                    // read by the NEXT launch only
    }
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws { size_t x, qkv, ctx, a, b, mid, logits, pos, total; };
Ws plan(const hirest_caption_decoder* d, int R) {
    Ws w; size_t off = 0;
    const size_t D = d->hidden;
    w.x = off; off += al((size_t)R * D * 4);
    w.qkv = off; off += al((size_t)R * 3 * D * 4);
    w.ctx = off; off += al((size_t)R * D * 4);
    w.a = off; off += al((size_t)R * D * 4);
    w.b = off; off += al((size_t)R * D * 4);
    w.mid = off; off += al((size_t)R * d->inter * 4);
    w.logits = off; off += al((size_t)R * d->vocab_padded * 4);
    w.pos = off; off += al((size_t)R * 4);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t hirest_caption_step_workspace_bytes(const hirest_caption_decoder* d, int32_t R) {
    if (!d || R <= 0) return 0;
    return plan(d, R).total;
}

#define CK(call) do { if (int e_ = (call)) return e_; } while (0)

extern "C" int hirest_caption_decode_step(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                                          const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                                          const float* const* enc_kv, int32_t F, const float* row_add, float* logp,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !d->layer || !last_ids || !kv_out || !enc_kv || !logp || !workspace || R <= 0 || F <= 0) return HIREST_E_BADARG;
    if (position < 0 || position >= d->max_pos || (position > 0 && (!kv_in || !parent_rows))) return HIREST_E_BADARG;
    const int D = d->hidden, H = d->heads;
    if (D % H != 0 || D / H != 64 || D % 4 != 0 || d->vocab_padded % 4 != 0) return HIREST_E_SHAPE;
    const Ws w = plan(d, R);
    if (workspace_bytes < w.total) return HIREST_E_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(base + w.x);
    float* qkv = reinterpret_cast<float*>(base + w.qkv);
    float* ctx = reinterpret_cast<float*>(base + w.ctx);
    float* a = reinterpret_cast<float*>(base + w.a);
    float* b = reinterpret_cast<float*>(base + w.b);
    float* mid = reinterpret_cast<float*>(base + w.mid);
    float* logits = reinterpret_cast<float*>(base + w.logits);
    int32_t* pos = reinterpret_cast<int32_t*>(base + w.pos);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float eps = 1e-12f, scale = 0.125f;            // BertLayerNorm eps; 1 / sqrt(64)

    hipLaunchKernelGGL(fill_i32_kernel, dim3((R + 255) / 256), dim3(256), 0, s, pos, R, position);
    CK(hirest_embedding_pos_fwd_f32(last_ids, pos, d->word_emb, d->pos_emb, a, R, D, stream));
    CK(hirest_layernorm(a, D, nullptr, d->emb_ln_g, d->emb_ln_b, eps, x, D, 1, R, D, stream));
    for (int i = 0; i < d->layers; ++i) {
        const hirest_caption_layer& L = d->layer[i];
        CK(hirest_gemm_f32(x, D, L.qkv_w, D, L.qkv_b, nullptr, 0, nullptr, 0, qkv, 3 * D, R, 3 * D, D, 0, stream));
        {
            const int64_t n4 = (int64_t)R * (position + 1) * (D / 4);
            hipLaunchKernelGGL(kv_gather_append_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                               position > 0 ? kv_in[2 * i] : nullptr, position > 0 ? kv_in[2 * i + 1] : nullptr,
                               position > 0 ? parent_rows : nullptr, qkv, kv_out[2 * i], kv_out[2 * i + 1], R, position, D);
        }
        // self-attention of the newest position (Tq = 1 per beam) over its position + 1 kept keys; nothing lies in the future
        CK(hirest_attention_f32_qkv(qkv, 3 * D, kv_out[2 * i], kv_out[2 * i + 1], D, ctx, R, 1, position + 1, H, 64, scale, 0.f, 0.f,
                                    stream));
        CK(hirest_gemm_f32(ctx, D, L.so_w, D, L.so_b, x, D, nullptr, 0, a, D, R, D, D, 0, stream));
        CK(hirest_layernorm(a, D, nullptr, L.so_ln_g, L.so_ln_b, eps, b, D, 1, R, D, stream));                  // s1 = b
        CK(hirest_gemm_f32(b, D, L.cq_w, D, L.cq_b, nullptr, 0, nullptr, 0, a, D, R, D, D, 0, stream));         // q2 = a
        CK(hirest_attention_f32_qkv(a, D, enc_kv[i], enc_kv[i] + D, 2 * D, ctx, R, 1, F, H, 64, scale, -10000.f, 0.f, stream));
        CK(hirest_gemm_f32(ctx, D, L.co_w, D, L.co_b, b, D, nullptr, 0, a, D, R, D, D, 0, stream));
        CK(hirest_layernorm(a, D, nullptr, L.co_ln_g, L.co_ln_b, eps, b, D, 1, R, D, stream));                  // d = b
        CK(hirest_gemm_f32(b, D, L.ff1_w, D, L.ff1_b, nullptr, 0, nullptr, 0, mid, d->inter, R, d->inter, D, 1, stream));
        CK(hirest_gemm_f32(mid, d->inter, L.ff2_w, d->inter, L.ff2_b, b, D, nullptr, 0, a, D, R, D, d->inter, 0, stream));
        CK(hirest_layernorm(a, D, nullptr, L.ff_ln_g, L.ff_ln_b, eps, x, D, 1, R, D, stream));
    }
    CK(hirest_gemm_f32(x, D, d->tr_w, D, d->tr_b, nullptr, 0, nullptr, 0, a, D, R, D, D, 1, stream));
    CK(hirest_layernorm(a, D, nullptr, d->tr_ln_g, d->tr_ln_b, eps, b, D, 1, R, D, stream));
    CK(hirest_gemm_f32(b, D, d->lm_w, D, d->lm_b, nullptr, 0, nullptr, 0, logits, d->vocab_padded, R, d->vocab_padded, D, 0, stream));
    CK(hirest_log_softmax_f32(logits, d->vocab_padded, row_add, logp, d->vocab_padded, R, d->vocab_padded, stream));
    return hirest_launch_status();
}

extern "C" int hirest_beam_advance(const float* val, const int32_t* idx, int32_t B, int32_t beam, int32_t vocab, int32_t step,
                                   int32_t max_steps, int32_t eos_id, float* scores, int32_t* tokens, int32_t* backptr, int32_t* n_steps,
                                   int32_t* done, int32_t* next_ids, int32_t* next_parents, float* next_add, void* stream) {
    if (!val || !idx || !scores || !tokens || !backptr || !n_steps || !done || !next_ids || !next_parents || !next_add) return HIREST_E_BADARG;
    if (B <= 0 || beam <= 0 || beam > 64 || vocab <= 0 || step < 0 || step >= max_steps) return HIREST_E_BADARG;
    hipLaunchKernelGGL(beam_advance_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), val, idx, beam, vocab, step,
                       max_steps, eos_id, scores, tokens, backptr, n_steps, done, next_ids, next_parents, next_add);
    return hirest_launch_status();
}
