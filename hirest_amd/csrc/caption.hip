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
#include <atomic>

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

// ---------------------------------------------------------------------------------------------
// The tail of a beam-search step in two kernels (it used to be log-softmax, chunked top-k, merge, beam bookkeeping and a copy of
// the done flags: 51 us of 270 per word).  Only the top `beam` of a sample's beam x vocabulary scores are ever read, so the
// log-probabilities are not materialised.  tail_scan, grid (nsel + 4, rows): blocks c < nsel pick the top `beam` LOGITS of their
// 4096-element chunk of a row (higher logit, then higher index: within a row the score is a monotone function of the logit);
// blocks nsel + g, g < 4, produce the row's max and four of the sixteen partial exp sums of log_softmax_kernel — that kernel's 1024
// threads are split over the four blocks thread for thread, so the log-sum-exp tail_select forms from them has the same bits.
// tail_select, one block per sample: score = ((x - max) - lse) + row_add of the beam x nsel x beam candidates, their top `beam`
// under topk_kernel's strict order (higher score, then higher flat index), then beam_advance_kernel's bookkeeping; the done flag
// also goes to pinned host memory when asked to.  Equal to the separate kernels unless two of a row's leading logits round to
// one score (the separate kernels then order the two by index, this one by logit).
// ---------------------------------------------------------------------------------------------
constexpr int TAIL_CHUNK = 4096, TAIL_STAT = 17;            // floats of statistics per row: 16 partial sums + the max
__device__ __forceinline__ bool tail_better(float sa, int ta, float sb, int tb) { return sa > sb || (sa == sb && ta > tb); }

// top k of (val, idx) pairs held 16 per thread by a 256-thread block; thread 0 reports pick j through put(j, val, idx).
// The pairs are packed into one 64-bit key each — the float's bits made monotone in the high word (-0 taken as +0), the index in the
// low word — so that topk_kernel's strict order (higher score, then higher index) is the unsigned order of the keys and a pick is a
// branch-free max: written with tail_better's && / || the compiler turned the 16-element scan into a cascade of ~50 exec-mask
// branches per pick (12 us for five picks).  One barrier per pick: the four wave winners go through a two-deep LDS buffer and
// every thread reduces them itself.
__device__ __forceinline__ unsigned long long tail_key(float v, int idx) {
    unsigned u = __builtin_bit_cast(unsigned, v + 0.0f);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return idx >= 0 ? ((unsigned long long)u << 32) | (unsigned)idx : 0ull;          // 0 = no element (a real key has a non-zero high word
}                                                                                      // unless the score is the most negative NaN pattern)
__device__ __forceinline__ float tail_key_value(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    return __builtin_bit_cast(float, u);
}
template <class Put>
__device__ __forceinline__ void block_topk16(const float (&val)[16], const int (&idx)[16], int k, Put put) {
    __shared__ unsigned long long rk[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long key[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) key[e] = tail_key(val[e], idx[e]);
    auto umax = [](unsigned long long a, unsigned long long b) { return a > b ? a : b; };
    unsigned long long prev = ~0ull;
    for (int j = 0; j < k; ++j) {
        unsigned long long best = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) best = umax(best, key[e] < prev ? key[e] : 0ull);
        {   // best of the wave on the cross-lane data paths (common.h: wave_max_x), the two words moving together
            int lo = (int)(unsigned)best, hi = (int)(unsigned)(best >> 32);
            auto join = [](int l, int h) { return ((unsigned long long)(unsigned)h << 32) | (unsigned)l; };
            int l2 = lo, h2 = hi;
            lane_swap32(lo, l2); lane_swap32(hi, h2); best = umax(join(lo, hi), join(l2, h2));
            lo = l2 = (int)(unsigned)best; hi = h2 = (int)(unsigned)(best >> 32);
            lane_swap16(lo, l2); lane_swap16(hi, h2); best = umax(join(lo, hi), join(l2, h2));
            lo = (int)(unsigned)best; hi = (int)(unsigned)(best >> 32);
            best = umax(best, join(lane_dpp<0x128>(lo), lane_dpp<0x128>(hi))); lo = (int)(unsigned)best; hi = (int)(unsigned)(best >> 32);
            best = umax(best, join(lane_dpp<0x124>(lo), lane_dpp<0x124>(hi))); lo = (int)(unsigned)best; hi = (int)(unsigned)(best >> 32);
            best = umax(best, join(lane_dpp<0x4E>(lo), lane_dpp<0x4E>(hi))); lo = (int)(unsigned)best; hi = (int)(unsigned)(best >> 32);
            best = umax(best, join(lane_dpp<0xB1>(lo), lane_dpp<0xB1>(hi)));
        }
        if (lane == 0) rk[j & 1][wave] = best;
        __syncthreads();
        best = umax(umax(rk[j & 1][0], rk[j & 1][1]), umax(rk[j & 1][2], rk[j & 1][3]));
        if (tid == 0) put(j, tail_key_value(best), best ? (int)(unsigned)best : -1);     // -1: fewer than j + 1 elements
        prev = best ? best : 0ull;                           // (nothing left: every later pick is empty too)
    }
}

__global__ __launch_bounds__(256) void tail_scan_kernel(const float* __restrict__ x, int64_t ldx, int V, int beam, int nsel,
                                                       float* __restrict__ cx, int32_t* __restrict__ ci, float* __restrict__ stat,
                                                       const float* __restrict__ tile_max, int ntile) {
    const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (int64_t)r * ldx;
    const int V4 = V >> 2;
    if (c < nsel) {
        float val[16];
        int idx[16];
        const int flat0 = (r % beam) * V;                   // flat index inside the sample's beam x vocabulary row
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = c * (TAIL_CHUNK / 4) + tid + 256 * u;
            const bool in = i < V4;                          // (loads are unconditional on a clamped index: a guarded load is a round trip of its own)
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * (in ? i : V4 - 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) { val[4 * u + e] = v[e]; idx[4 * u + e] = in ? flat0 + 4 * i + e : -1; }
        }
        const int64_t ob = ((int64_t)r * nsel + c) * beam;
        block_topk16(val, idx, beam, [&](int j, float sv, int iv) { cx[ob + j] = sv; ci[ob + j] = iv; });
        return;
    }
    // statistics block g: log_softmax_kernel's thread T = 256 g + tid (wave 4 g + wave) adds vectors T, T + 1024, ...
    __shared__ float red[4];
    __shared__ float bc;
    const int g = c - nsel;
    constexpr int NM = 32;                                   // vectors per thread for the row max: V <= 32768 (launcher)
    float mx = -INFINITY;
    f32x4 mine[8], all[NM];                                  // every load unconditional (clamped index) and issued before the first use
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = tid + 256 * g + 1024 * j;
        mine[j] = *reinterpret_cast<const f32x4*>(xr + 4 * (i < V4 ? i : V4 - 1));
    }
    if (tile_max) {                                          // (uniform) the LM head reported the maximum of every 16-column tile of the row:
        float tm[8];                                         // ntile floats instead of the whole row (7.6 KB instead of 122)
#pragma unroll
        for (int m = 0; m < 8; ++m) { const int i = tid + 256 * m; tm[m] = tile_max[(int64_t)r * ntile + (i < ntile ? i : ntile - 1)]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) mx = fmaxf(mx, tm[m]);
    } else {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int i = tid + 256 * m;
            all[m] = *reinterpret_cast<const f32x4*>(xr + 4 * (i < V4 ? i : V4 - 1));
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) mx = fmaxf(fmaxf(mx, fmaxf(all[m][0], all[m][1])), fmaxf(all[m][2], all[m][3]));   // (a repeated element changes no max)
    }
    mx = wave_max_x(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = bc;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (tid + 256 * g + 1024 * j < V4)
            s += (expf(mine[j][0] - mx) + expf(mine[j][1] - mx)) + (expf(mine[j][2] - mx) + expf(mine[j][3] - mx));
    s = wave_sum_x(s);
    if (lane == 0) stat[(int64_t)r * TAIL_STAT + 4 * g + wave] = s;
    if (g == 0 && tid == 0) stat[(int64_t)r * TAIL_STAT + 16] = mx;
}

// one block per sample: scores of its beam x nsel x beam candidates, their top `beam`, then beam_advance_kernel's bookkeeping
__global__ __launch_bounds__(256) void tail_select_kernel(const float* __restrict__ cx, const int32_t* __restrict__ ci,
                                                         const float* __restrict__ stat, const float* row_add, int nsel,
                                                         int beam, int vocab, int step, int max_steps, int eos, float* __restrict__ scores,
                                                         int32_t* __restrict__ tokens, int32_t* __restrict__ backptr,
                                                         int32_t* __restrict__ n_steps, int32_t* __restrict__ done,
                                                         int32_t* __restrict__ next_ids, int32_t* __restrict__ next_parents,
                                                         float* next_add, int32_t* __restrict__ done_host) {
    // row_add and next_add MAY ALIAS (hirest_caption_beam_step passes one buffer: this step's running scores in, the next step's
    // out), hence no __restrict__ on either; every read of row_add happens before the block's first store to next_add.
    __shared__ float top_s[16];
    __shared__ int top_i[16];
    __shared__ float row_mx[16], row_lse[16], row_ad[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    // every load of the kernel is issued here, unconditionally and on clamped indices, before the first use: three dependent round
    // trips (done, the row statistics, the candidates) become one
    const int per_row = nsel * beam, ncand = beam * per_row;  // <= 16 * 8 * 16 = 2048 = 8 per thread; 16 slots
    const int was_done = done[b];                            // (block-uniform: written only by this block, in an earlier launch)
    const int rb = b * beam + (tid < beam ? tid : beam - 1);
    float stv[TAIL_STAT];
#pragma unroll
    for (int w = 0; w < TAIL_STAT; ++w) stv[w] = stat[(int64_t)rb * TAIL_STAT + w];
    const float radd = row_add[rb];
    float val[16];
    int idx[16];
    int civ[16];
    float cxv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int v = tid + 256 * u, vc = v < ncand ? v : ncand - 1;
        civ[u] = ci[(int64_t)b * ncand + vc];
        cxv[u] = cx[(int64_t)b * ncand + vc];
    }
    if (was_done != 0) {
        if (tid < beam) { const int row = b * beam + tid; next_ids[row] = eos; next_parents[row] = row; next_add[row] = 0.f; }
        if (tid == 0 && done_host) done_host[b] = ((step + 1) << 1) | 1;
        return;
    }
    if (tid < beam) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += stv[w];           // log_softmax_kernel's fixed order
        row_lse[tid] = logf(t); row_mx[tid] = stv[16]; row_ad[tid] = radd;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int v = tid + 256 * u, rr = (v < ncand ? v : ncand - 1) / per_row;
        val[u] = ((cxv[u] - row_mx[rr]) - row_lse[rr]) + row_ad[rr];
        idx[u] = v < ncand ? civ[u] : -1;
    }
    block_topk16(val, idx, beam, [&](int j, float sv, int iv) { top_s[j] = sv; top_i[j] = iv; });
    if (tid < beam) {                                        // beam_advance_kernel, thread k = beam k
        const int k = tid, row = b * beam + k;
        const int flat = top_i[k];
        const int prev = flat / vocab, word = flat - prev * vocab;
        const float v = top_s[k];
        scores[row] = v;
        tokens[((int64_t)b * max_steps + step) * beam + k] = word;
        backptr[((int64_t)b * max_steps + step) * beam + k] = prev;
        next_ids[row] = word; next_parents[row] = b * beam + prev; next_add[row] = v;
        if (k == 0) {
            n_steps[b] = step + 1;
            const int fin = word == eos ? 1 : 0;
            if (fin) done[b] = 1;                            // read by the NEXT launch only
            if (done_host) done_host[b] = ((step + 1) << 1) | fin;   // stamped: the host polls the word itself, no event per step
        }
    }
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws { size_t x, qkv, ctx, a, b, mid, logits, pos, tm, total; };
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
    w.tm = off; off += al((size_t)R * ((d->vocab_padded + 15) / 16) * 4);       // the LM head's tile maxima (hirest_caption_beam_step)
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t hirest_caption_step_workspace_bytes(const hirest_caption_decoder* d, int32_t R) {
    if (!d || R <= 0) return 0;
    return plan(d, R).total;
}

#define CK(call) do { if (int e_ = (call)) return e_; } while (0)

static std::atomic<int> g_caption_mode{0};     // hirest_caption_select: A/B and tests
// 0 = LayerNorms / embedding inside the GEMMs and one-query attention kernels, 1 = separate LayerNorm / embedding / K-V gather kernels
extern "C" int hirest_caption_select(int32_t mode) {
    if (mode < 0 || mode > 1) return HIREST_E_BADARG;
    g_caption_mode = mode;
    return 0;
}

// logp != NULL: log_softmax + row_add into logp; logits_out != NULL: the raw LM-head logits there instead (hirest_caption_beam_tail)
static int decode_step(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                       const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                       const float* const* enc_kv, int32_t F, const float* row_add, float* logp, float* logits_out, float** tile_max_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !d->layer || !last_ids || !kv_out || !enc_kv || (!logp && !logits_out) || !workspace || R <= 0 || F <= 0) return HIREST_E_BADARG;
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
    float* logits = logits_out ? logits_out : reinterpret_cast<float*>(base + w.logits);
    int32_t* pos = reinterpret_cast<int32_t*>(base + w.pos);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float eps = 1e-12f, scale = 0.125f;            // BertLayerNorm eps; 1 / sqrt(64)

    // (the LM head's LayerNorm-prologue form exists as the streaming kernel for D = 768 only: hirest_gemm_f32_ln rejects N >= 8192
    //  with another depth, so such a decoder takes the separate-kernel path instead of failing the step)
    // (above 256 rows the LayerNorm-GEMMs exist in the row-group streaming form only, i.e. for D = 768)
    const bool fused_ln = g_caption_mode == 0 && (R <= 256 || D == 768) && D % 256 == 0 && D <= 1024 && (d->vocab_padded < 8192 || D == 768);
    if (fused_ln) {
        // every LayerNorm (and the token + position embedding) is the prologue of the GEMM that consumes it (hirest_gemm_f32_ln):
        // a = the pre-LayerNorm sum of the previous sub-layer, x / b = the normalised rows (written by the GEMM, residual of the next)
        const float* pos_row = d->pos_emb + (int64_t)position * D;
        float* q2 = qkv;                                              // [R, D]: the packed q | k | v rows are dead after the self-attention
        for (int i = 0; i < d->layers; ++i) {
            const hirest_caption_layer& L = d->layer[i];
            if (i == 0) CK(hirest_gemm_f32_ln(nullptr, 0, last_ids, d->word_emb, pos_row, d->emb_ln_g, d->emb_ln_b, eps, x, D, L.qkv_w, D, L.qkv_b,
                                              nullptr, 0, qkv, 3 * D, R, 3 * D, D, 0, stream));
            else CK(hirest_gemm_f32_ln(a, D, nullptr, nullptr, nullptr, d->layer[i - 1].ff_ln_g, d->layer[i - 1].ff_ln_b, eps, x, D, L.qkv_w, D,
                                       L.qkv_b, nullptr, 0, qkv, 3 * D, R, 3 * D, D, 0, stream));
            // self-attention of the newest position over the parent beam's kept keys + its own, which also writes this beam's
            // history for the next step (one wave per (row, head): hirest_attention_f32_decode)
            CK(hirest_attention_f32_decode(qkv, 3 * D, position > 0 ? kv_in[2 * i] : nullptr, position > 0 ? kv_in[2 * i + 1] : nullptr, D,
                                           position > 0 ? parent_rows : nullptr, position, qkv + D, qkv + 2 * D, 3 * D, kv_out[2 * i],
                                           kv_out[2 * i + 1], ctx, R, H, scale, 0.f, 0.f, stream));
            CK(hirest_gemm_f32(ctx, D, L.so_w, D, L.so_b, x, D, nullptr, 0, a, D, R, D, D, 0, stream));
            CK(hirest_gemm_f32_ln(a, D, nullptr, nullptr, nullptr, L.so_ln_g, L.so_ln_b, eps, b, D, L.cq_w, D, L.cq_b, nullptr, 0, q2, D, R, D, D,
                                  0, stream));                                                                  // s1 = b
            CK(hirest_attention_f32_decode(q2, D, enc_kv[i], enc_kv[i] + D, 2 * D, nullptr, F, nullptr, nullptr, 0, nullptr, nullptr, ctx, R, H,
                                           scale, -10000.f, 0.f, stream));
            CK(hirest_gemm_f32(ctx, D, L.co_w, D, L.co_b, b, D, nullptr, 0, a, D, R, D, D, 0, stream));
            CK(hirest_gemm_f32_ln(a, D, nullptr, nullptr, nullptr, L.co_ln_g, L.co_ln_b, eps, b, D, L.ff1_w, D, L.ff1_b, nullptr, 0, mid,
                                  d->inter, R, d->inter, D, 1, stream));                                        // d = b
            CK(hirest_gemm_f32(mid, d->inter, L.ff2_w, d->inter, L.ff2_b, b, D, nullptr, 0, a, D, R, D, d->inter, 0, stream));
        }
        const hirest_caption_layer& Z = d->layer[d->layers - 1];
        CK(hirest_gemm_f32_ln(a, D, nullptr, nullptr, nullptr, Z.ff_ln_g, Z.ff_ln_b, eps, nullptr, 0, d->tr_w, D, d->tr_b, nullptr, 0, x, D, R, D,
                              D, 1, stream));
        // LM head with the transform's LayerNorm as its prologue (persistent blocks: the rows are normalised once per CU); for the
        // one-call beam step it also leaves the maxima of its 16-column tiles for the tail
        if (R >= 64 && d->vocab_padded >= 8192 && d->lm_w2 && D % 32 == 0) {
            // split-operand LM head (MomentModel.set_precision('bf16x3')): LayerNorm straight into the split format (the dead `mid` rows hold
            // it), then the 128 x 128 split-operand GEMM over [R, vocab]; the tail scans the rows itself (no tile maxima from this kernel)
            hirest_bf16* b2 = reinterpret_cast<hirest_bf16*>(mid);
            CK(hirest_layernorm_f32_split2(x, D, nullptr, 0, d->tr_ln_g, d->tr_ln_b, eps, nullptr, 0, b2, 2 * D, R, D, stream));
            hirest_gemm_args g;
            g.struct_size = sizeof(g);
            g.A = b2; g.lda = 2 * D; g.W = d->lm_w2; g.ldw = 2 * D; g.bias = d->lm_b; g.out = logits; g.ldo = d->vocab_padded;
            g.M = R; g.N = d->vocab_padded; g.K = 2 * D; g.epilogue = HIREST_EPI_BIAS_F32; g.pos = nullptr; g.patches_per_frame = 0;
            g.aux0 = g.aux1 = nullptr; g.flags = HIREST_GEMM_X3 | HIREST_GEMM_X3_T128;
            CK(hirest_gemm_bf16(&g, stream));
        } else if (R > 32 && d->vocab_padded >= 8192) {
            // a merged search (60 - 160 beam rows): by now a compute problem — LayerNorm once, then the row-group streaming product
            // with five row tiles per wave (its LayerNorm-prologue form holds three), which also leaves the tile maxima for the tail
            CK(hirest_layernorm(x, D, nullptr, d->tr_ln_g, d->tr_ln_b, eps, b, D, 1, R, D, stream));
            if (tile_max_out && D == 768 && hirest_gemm_f32_rows_preferred(R)) {
                float* tm = reinterpret_cast<float*>(base + w.tm);
                CK(hirest_gemm_f32_rows_colmax(b, D, d->lm_w, D, d->lm_b, logits, d->vocab_padded, tm, R, d->vocab_padded, D, stream));
                *tile_max_out = tm;
            } else {
                CK(hirest_gemm_f32(b, D, d->lm_w, D, d->lm_b, nullptr, 0, nullptr, 0, logits, d->vocab_padded, R, d->vocab_padded, D, 0, stream));
            }
        } else if (tile_max_out && d->vocab_padded >= 8192 && D == 768) {
            float* tm = reinterpret_cast<float*>(base + w.tm);
            CK(hirest_gemm_f32_ln_colmax(x, D, d->tr_ln_g, d->tr_ln_b, eps, d->lm_w, D, d->lm_b, logits, d->vocab_padded, tm, R, d->vocab_padded, D,
                                         stream));
            *tile_max_out = tm;
        } else {
            CK(hirest_gemm_f32_ln(x, D, nullptr, nullptr, nullptr, d->tr_ln_g, d->tr_ln_b, eps, nullptr, 0, d->lm_w, D, d->lm_b, nullptr, 0,
                                  logits, d->vocab_padded, R, d->vocab_padded, D, 0, stream));
        }
        if (logp) CK(hirest_log_softmax_f32(logits, d->vocab_padded, row_add, logp, d->vocab_padded, R, d->vocab_padded, stream));
        return hirest_launch_status();
    } else {
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
    }
    CK(hirest_gemm_f32(b, D, d->lm_w, D, d->lm_b, nullptr, 0, nullptr, 0, logits, d->vocab_padded, R, d->vocab_padded, D, 0, stream));
    if (logp) CK(hirest_log_softmax_f32(logits, d->vocab_padded, row_add, logp, d->vocab_padded, R, d->vocab_padded, stream));
    return hirest_launch_status();
}

extern "C" int hirest_caption_decode_step(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                                          const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                                          const float* const* enc_kv, int32_t F, const float* row_add, float* logp,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!logp) return HIREST_E_BADARG;
    return decode_step(d, R, position, last_ids, parent_rows, kv_in, kv_out, enc_kv, F, row_add, logp, nullptr, nullptr, workspace,
                       workspace_bytes, stream);
}

extern "C" int hirest_caption_decode_logits(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                                            const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                                            const float* const* enc_kv, int32_t F, float* logits, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    if (!logits) return HIREST_E_BADARG;
    return decode_step(d, R, position, last_ids, parent_rows, kv_in, kv_out, enc_kv, F, nullptr, nullptr, logits, nullptr, workspace,
                       workspace_bytes, stream);
}

extern "C" size_t hirest_caption_beam_tail_workspace_bytes(int32_t B, int32_t beam, int32_t vocab) {
    if (B <= 0 || beam <= 0 || vocab <= 0) return 0;
    const size_t nsel = ((size_t)vocab + TAIL_CHUNK - 1) / TAIL_CHUNK, R = (size_t)B * beam;
    return R * nsel * beam * 8 + R * TAIL_STAT * 4;
}

static int beam_tail(const float* logits, int64_t ldx, const float* row_add, const float* tile_max, int32_t B, int32_t beam, int32_t vocab,
                     int32_t step, int32_t max_steps, int32_t eos_id, float* scores, int32_t* tokens,
                     int32_t* backptr, int32_t* n_steps, int32_t* done, int32_t* next_ids, int32_t* next_parents,
                     float* next_add, int32_t* done_host, void* workspace, size_t workspace_bytes, void* stream) {
    if (!logits || !row_add || !scores || !tokens || !backptr || !n_steps || !done || !next_ids || !next_parents || !next_add || !workspace)
        return HIREST_E_BADARG;
    if (B <= 0 || beam <= 0 || beam > 16 || vocab <= 0 || step < 0 || step >= max_steps) return HIREST_E_BADARG;
    if (vocab % 4 != 0 || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(logits) & 15) != 0 || beam > vocab || vocab > 32768) return HIREST_E_SHAPE;
    if (workspace_bytes < hirest_caption_beam_tail_workspace_bytes(B, beam, vocab)) return HIREST_E_WORKSPACE;
    const int nsel = (vocab + TAIL_CHUNK - 1) / TAIL_CHUNK;
    const int64_t n = (int64_t)B * beam * nsel * beam;
    float* cx = reinterpret_cast<float*>(workspace);
    int32_t* ci = reinterpret_cast<int32_t*>(cx + n);
    float* stat = reinterpret_cast<float*>(ci + n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ntile = (vocab + 15) / 16;
    hipLaunchKernelGGL(tail_scan_kernel, dim3(nsel + 4, B * beam), dim3(256), 0, s, logits, ldx, vocab, beam, nsel, cx, ci, stat,
                       ntile <= 2048 ? tile_max : nullptr, ntile);
    hipLaunchKernelGGL(tail_select_kernel, dim3(B), dim3(256), 0, s, cx, ci, stat, row_add, nsel, beam, vocab, step, max_steps, eos_id,
                       scores, tokens, backptr, n_steps, done, next_ids, next_parents, next_add, done_host);
    return hirest_launch_status();
}
extern "C" int hirest_caption_beam_tail(const float* logits, int64_t ldx, const float* row_add, int32_t B, int32_t beam, int32_t vocab,
                                        int32_t step, int32_t max_steps, int32_t eos_id, float* scores, int32_t* tokens,
                                        int32_t* backptr, int32_t* n_steps, int32_t* done, int32_t* next_ids, int32_t* next_parents,
                                        float* next_add, int32_t* done_host, void* workspace, size_t workspace_bytes, void* stream) {
    return beam_tail(logits, ldx, row_add, nullptr, B, beam, vocab, step, max_steps, eos_id, scores, tokens, backptr, n_steps, done, next_ids,
                     next_parents, next_add, done_host, workspace, workspace_bytes, stream);
}

// One word of the beam search in one call: hirest_caption_decode_logits, then the tail on its logits — with the LM head's per-tile
// maxima (kept in the decode workspace) in place of the tail's row scan when the step ran the fused kernels.
extern "C" int hirest_caption_beam_step(const hirest_caption_decoder* d, int32_t B, int32_t beam, int32_t position, int32_t* ids,
                                        int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                                        const float* const* enc_kv, int32_t F, float* row_add, float* logits, int32_t max_steps,
                                        int32_t eos_id, float* scores, int32_t* tokens, int32_t* backptr, int32_t* n_steps, int32_t* done,
                                        int32_t* done_host, void* workspace, size_t workspace_bytes, void* tail_workspace,
                                        size_t tail_workspace_bytes, void* stream) {
    if (!d || B <= 0 || beam <= 0 || !logits) return HIREST_E_BADARG;
    const int R = B * beam;
    float* tm = nullptr;
    if (int e = decode_step(d, R, position, ids, position > 0 ? parent_rows : nullptr, kv_in, kv_out, enc_kv, F, nullptr, nullptr, logits, &tm,
                            workspace, workspace_bytes, stream))
        return e;
    return beam_tail(logits, d->vocab_padded, row_add, tm, B, beam, d->vocab_padded, position, max_steps, eos_id, scores, tokens, backptr,
                     n_steps, done, ids, parent_rows, row_add, done_host, tail_workspace, tail_workspace_bytes, stream);
}

// collect_hypothesis_and_scores(..., n_best = 1) (clip4caption/train.py:590-599, beam.py:31-123) on the device: per sample the best beam (highest
// score, lowest beam number among equals — torch.sort's order on a sorted top-k) walked back through the recorded parents.  out[b] = n | the n words.
__global__ void beam_backtrack_kernel(const float* __restrict__ scores, const int32_t* __restrict__ tokens, const int32_t* __restrict__ backptr,
                                      const int32_t* __restrict__ n_steps, int beam, int max_steps, int B, int32_t* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int k = 0;
    float best = scores[(int64_t)b * beam];
    for (int i = 1; i < beam; ++i) { const float v = scores[(int64_t)b * beam + i]; if (v > best) { best = v; k = i; } }
    const int n = n_steps[b] < max_steps ? n_steps[b] : max_steps;
    int32_t* o = out + (int64_t)b * (max_steps + 1);
    o[0] = n;
    for (int j = n - 1; j >= 0; --j) {
        const int64_t at = ((int64_t)b * max_steps + j) * beam + k;
        o[1 + j] = tokens[at];
        k = backptr[at];
    }
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

extern "C" int hirest_beam_backtrack(const float* scores, const int32_t* tokens, const int32_t* backptr, const int32_t* n_steps, int32_t B,
                                     int32_t beam, int32_t max_steps, int32_t* out, void* stream) {
    if (!scores || !tokens || !backptr || !n_steps || !out || B <= 0 || beam <= 0 || max_steps <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3((B + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), scores, tokens, backptr,
                       n_steps, beam, max_steps, B, out);
    return hirest_launch_status();
}
