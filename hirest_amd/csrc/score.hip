// Pooling + scoring: frame -> video mean pooling with L2 normalisation, the text x video cosine
// matrix in fp32, and per-query top-k with the reference's tie rule
// (inference_video_retrieval.py:283-285,323-334; evaluate.py:58-60).  All HBM-bound / tiny.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// one block per video
__global__ __launch_bounds__(256) void pool_l2_kernel(const float* __restrict__ fe, float* __restrict__ out, int F, int E,
                                                     int norm_first, const int32_t* __restrict__ seg_off) {
    extern __shared__ float sm[];          // [F] per-frame norms, then [8] reduction scratch
    float* fnorm = sm;
    float* red = sm + (seg_off ? 0 : F);
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = fe + (int64_t)v * F * E;
    if (seg_off) {                         // ragged segments of a packed [rows, E] matrix (never with norm_first)
        base = fe + (int64_t)seg_off[v] * E;
        F = seg_off[v + 1] - seg_off[v];
        if (F <= 0) {                      // an empty segment has no mean: zeros
            for (int c = tid; c < E; c += 256) out[(int64_t)v * E + c] = 0.f;
            return;
        }
    }
    if (norm_first) {
        for (int f = wave; f < F; f += 4) {
            float s = 0.f;
            for (int c = lane; c < (E >> 2); c += 64) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(base + (int64_t)f * E + 4 * c);
                s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            }
            s = wave_sum(s);
            if (lane == 0) fnorm[f] = sqrtf(s);
        }
        __syncthreads();
    }
    float ss = 0.f;
    for (int c = tid; c < (E >> 2); c += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < F; ++f) {
            f32x4 x = *reinterpret_cast<const f32x4*>(base + (int64_t)f * E + 4 * c);
            if (norm_first) { const float n = fnorm[f]; x[0] /= n; x[1] /= n; x[2] /= n; x[3] /= n; }
            acc += x;
        }
        const float fF = (float)F;
        acc[0] /= fF; acc[1] /= fF; acc[2] /= fF; acc[3] /= fF;
        *reinterpret_cast<f32x4*>(out + (int64_t)v * E + 4 * c) = acc;   // un-normalised mean, fixed up below
        ss += acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
    }
    const float nrm = sqrtf(block_sum(ss, red));
    for (int c = tid; c < (E >> 2); c += 256) {
        f32x4 a = *reinterpret_cast<f32x4*>(out + (int64_t)v * E + 4 * c);   // same thread wrote it
        a[0] /= nrm; a[1] /= nrm; a[2] /= nrm; a[3] /= nrm;
        *reinterpret_cast<f32x4*>(out + (int64_t)v * E + 4 * c) = a;
    }
}

// scores[q][v] = sum_e T[q][e] V[v][e]: 64x64 tile per block, 4x4 per thread, fp32 FMA.
__global__ __launch_bounds__(256) void similarity_kernel(const float* __restrict__ T, const float* __restrict__ Vn,
                                                        float* __restrict__ S, int Q, int V, int E) {
    __shared__ float ts[16][65];
    __shared__ float vs[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int q0 = blockIdx.y * 64, v0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int e0 = 0; e0 < E; e0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, c = i & 15;
            const int q = q0 + r, v = v0 + r;
            ts[c][r] = (q < Q && e0 + c < E) ? T[(int64_t)q * E + e0 + c] : 0.f;
            vs[c][r] = (v < V && e0 + c < E) ? Vn[(int64_t)v * E + e0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = ts[c][ty * 4 + i]; b[i] = vs[c][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + ty * 4 + i, v = v0 + tx * 4 + j;
            if (q < Q && v < V) S[(int64_t)q * V + v] = acc[i][j];
        }
}

// strict total order used by evaluate.py: higher score first, ties by higher tie key first
__device__ __forceinline__ bool better(float sa, int ta, float sb, int tb) { return sa > sb || (sa == sb && ta > tb); }

// one block per query; k selection passes, each finds the best element strictly below the last pick
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ scores, const int32_t* __restrict__ tie_rank,
                                                  int V, int k, int32_t* __restrict__ out_index, float* __restrict__ out_score) {
    __shared__ float rs[4];
    __shared__ int rt[4];
    __shared__ int ri[4];
    __shared__ float ps; __shared__ int pt;
    const float* row = scores + (int64_t)blockIdx.x * V;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float prev_s = INFINITY; int prev_t = 0x7fffffff;
    for (int j = 0; j < k; ++j) {
        float bs = -INFINITY; int bt = -0x7fffffff - 1; int bi = -1;
        for (int v = tid; v < V; v += 256) {
            const float s = row[v];
            const int t = tie_rank ? tie_rank[v] : v;
            if (better(prev_s, prev_t, s, t) && (bi < 0 || better(s, t, bs, bt))) { bs = s; bt = t; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int ot = __shfl_xor(bt, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi >= 0 && (bi < 0 || better(os, ot, bs, bt))) { bs = os; bt = ot; bi = oi; }
        }
        if (lane == 0) { rs[wave] = bs; rt[wave] = bt; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float s = rs[0]; int t = rt[0]; int i = ri[0];
            for (int w = 1; w < 4; ++w)
                if (ri[w] >= 0 && (i < 0 || better(rs[w], rt[w], s, t))) { s = rs[w]; t = rt[w]; i = ri[w]; }
            out_index[(int64_t)blockIdx.x * k + j] = i;
            if (out_score) out_score[(int64_t)blockIdx.x * k + j] = s;
            ps = s; pt = t;
        }
        __syncthreads();
        prev_s = ps; prev_t = pt;
        __syncthreads();
    }
}

// Long rows (beam search: 5 rows x beams*vocab = 152 620 scores) as two passes: every 4096-element chunk of a row yields
// its own top-k candidates (score, tie key, index) in parallel, then one block per row selects among the candidates.
// Same strict order, so the result equals topk_kernel's.
constexpr int TOPK_CHUNK = 4096;
__global__ __launch_bounds__(256) void topk_chunk_kernel(const float* __restrict__ scores, const int32_t* __restrict__ tie_rank,
                                                        int V, int k, int nchunk, float* __restrict__ cs, int32_t* __restrict__ ct,
                                                        int32_t* __restrict__ ci) {
    __shared__ float rs[4];
    __shared__ int rt[4];
    __shared__ int ri[4];
    __shared__ float ps; __shared__ int pt;
    const int q = blockIdx.y, c = blockIdx.x;
    const float* row = scores + (int64_t)q * V;
    const int v0 = c * TOPK_CHUNK, v1 = min(V, v0 + TOPK_CHUNK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float prev_s = INFINITY; int prev_t = 0x7fffffff;
    const int64_t ob = ((int64_t)q * nchunk + c) * k;
    for (int j = 0; j < k; ++j) {
        float bs = -INFINITY; int bt = -0x7fffffff - 1; int bi = -1;
        for (int v = v0 + tid; v < v1; v += 256) {
            const float s = row[v];
            const int t = tie_rank ? tie_rank[v] : v;
            if (better(prev_s, prev_t, s, t) && (bi < 0 || better(s, t, bs, bt))) { bs = s; bt = t; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int ot = __shfl_xor(bt, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi >= 0 && (bi < 0 || better(os, ot, bs, bt))) { bs = os; bt = ot; bi = oi; }
        }
        if (lane == 0) { rs[wave] = bs; rt[wave] = bt; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float s = rs[0]; int t = rt[0]; int i = ri[0];
            for (int w = 1; w < 4; ++w)
                if (ri[w] >= 0 && (i < 0 || better(rs[w], rt[w], s, t))) { s = rs[w]; t = rt[w]; i = ri[w]; }
            cs[ob + j] = s; ct[ob + j] = t; ci[ob + j] = i;          // i < 0: the chunk has fewer than j+1 elements
            ps = s; pt = t;
        }
        __syncthreads();
        prev_s = ps; prev_t = pt;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ct,
                                                        const int32_t* __restrict__ ci, int ncand, int k,
                                                        int32_t* __restrict__ out_index, float* __restrict__ out_score) {
    __shared__ float rs[4];
    __shared__ int rt[4];
    __shared__ int ri[4];
    __shared__ float ps; __shared__ int pt;
    const int q = blockIdx.x;
    const float* s_ = cs + (int64_t)q * ncand;
    const int32_t* t_ = ct + (int64_t)q * ncand;
    const int32_t* i_ = ci + (int64_t)q * ncand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float prev_s = INFINITY; int prev_t = 0x7fffffff;
    for (int j = 0; j < k; ++j) {
        float bs = -INFINITY; int bt = -0x7fffffff - 1; int bi = -1;
        for (int v = tid; v < ncand; v += 256) {
            const int idx = i_[v];
            if (idx < 0) continue;
            const float s = s_[v];
            const int t = t_[v];
            if (better(prev_s, prev_t, s, t) && (bi < 0 || better(s, t, bs, bt))) { bs = s; bt = t; bi = idx; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(bs, o, 64);
            const int ot = __shfl_xor(bt, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi >= 0 && (bi < 0 || better(os, ot, bs, bt))) { bs = os; bt = ot; bi = oi; }
        }
        if (lane == 0) { rs[wave] = bs; rt[wave] = bt; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float s = rs[0]; int t = rt[0]; int i = ri[0];
            for (int w = 1; w < 4; ++w)
                if (ri[w] >= 0 && (i < 0 || better(rs[w], rt[w], s, t))) { s = rs[w]; t = rt[w]; i = ri[w]; }
            out_index[(int64_t)q * k + j] = i;
            if (out_score) out_score[(int64_t)q * k + j] = s;
            ps = s; pt = t;
        }
        __syncthreads();
        prev_s = ps; prev_t = pt;
        __syncthreads();
    }
}

}  // namespace

extern "C" int hirest_pool_l2norm(const float* frame_embeds, float* out, int32_t V, int32_t F, int32_t E,
                                  int32_t normalize_frames_first, void* stream) {
    if (!frame_embeds || !out || V <= 0 || F <= 0 || E <= 0) return HIREST_E_BADARG;
    if (E % 4 != 0 || F > 8192) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pool_l2_kernel, dim3(V), dim3(256), (F + 8) * sizeof(float), s, frame_embeds, out, F, E,
                       normalize_frames_first, nullptr);
    return hirest_launch_status();
}

extern "C" int hirest_pool_l2norm_varlen(const float* rows, const int32_t* seg_off, float* out, int32_t V, int32_t E, void* stream) {
    if (!rows || !seg_off || !out || V <= 0 || E <= 0) return HIREST_E_BADARG;
    if (E % 4 != 0) return HIREST_E_SHAPE;
    hipLaunchKernelGGL(pool_l2_kernel, dim3(V), dim3(256), 8 * sizeof(float), reinterpret_cast<hipStream_t>(stream), rows, out, 0, E, 0,
                       seg_off);
    return hirest_launch_status();
}

extern "C" int hirest_similarity_f32(const float* text_n, const float* video_n, float* scores, int32_t Q, int32_t V,
                                     int32_t E, void* stream) {
    if (!text_n || !video_n || !scores || Q <= 0 || V <= 0 || E <= 0) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(similarity_kernel, dim3((V + 63) / 64, (Q + 63) / 64), dim3(256), 0, s, text_n, video_n, scores, Q, V, E);
    return hirest_launch_status();
}

extern "C" int hirest_topk_f32(const float* scores, const int32_t* tie_rank, int32_t Q, int32_t V, int32_t k,
                               int32_t* out_index, float* out_score, void* stream) {
    if (!scores || !out_index || Q <= 0 || V <= 0 || k <= 0 || k > V) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(topk_kernel, dim3(Q), dim3(256), 0, s, scores, tie_rank, V, k, out_index, out_score);
    return hirest_launch_status();
}

extern "C" int64_t hirest_topk_workspace_bytes(int32_t Q, int32_t V, int32_t k) {
    if (Q <= 0 || V <= 0 || k <= 0) return HIREST_E_BADARG;
    const int64_t nchunk = (V + TOPK_CHUNK - 1) / TOPK_CHUNK;
    return Q * nchunk * k * 12;
}

extern "C" int hirest_topk_f32_ws(const float* scores, const int32_t* tie_rank, int32_t Q, int32_t V, int32_t k,
                                  int32_t* out_index, float* out_score, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!scores || !out_index || Q <= 0 || V <= 0 || k <= 0 || k > V) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = (V + TOPK_CHUNK - 1) / TOPK_CHUNK;
    if (nchunk < 4 || Q > 65535) {                                         // short rows: one pass is enough
        hipLaunchKernelGGL(topk_kernel, dim3(Q), dim3(256), 0, s, scores, tie_rank, V, k, out_index, out_score);
        return hirest_launch_status();
    }
    const int64_t n = (int64_t)Q * nchunk * k;
    if (!workspace || workspace_bytes < n * 12) return HIREST_E_WORKSPACE;
    float* cs = reinterpret_cast<float*>(workspace);
    int32_t* ct = reinterpret_cast<int32_t*>(cs + n);
    int32_t* ci = ct + n;
    hipLaunchKernelGGL(topk_chunk_kernel, dim3(nchunk, Q), dim3(256), 0, s, scores, tie_rank, V, k, nchunk, cs, ct, ci);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(Q), dim3(256), 0, s, cs, ct, ci, nchunk * k, k, out_index, out_score);
    return hirest_launch_status();
}
