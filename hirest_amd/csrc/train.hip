// Training side of the joint model: the backward pass of MomentModel.train_moment_retrieval
// (/root/reference/modeling.py:155-270: fusion -> VisualModel -> start / end heads -> masked BCE) for the 63 M trainable
// parameters, in exact fp32 like the forward kernels of joint.hip.  Matrix products reuse hirest_gemm_f32 on transposed
// operands (dX = dY W: A = dY, "W" = W^T;  dW = dY^T X: A = dY^T, "W" = X^T, reduction over the zero-padded row count); this
// file holds what is not a GEMM: transposes, (weighted / selected) column sums for bias, LayerNorm-affine, embedding-table and
// head-weight gradients, LayerNorm / GELU / tanh backward, the attention forward that keeps its probabilities and its
// backward, dropout, the fusion's elementwise backward and the loss.  These problems are small (B*T <= a few thousand rows):
// one wave per row, no tiling heroics.
#include "common.h"
#include <atomic>

namespace {

// counter-based keep mask of the dropout sites: a pure function of (seed, element index), so backward regenerates it
__device__ __forceinline__ float keep_scale(uint32_t seed, uint64_t idx, float p) {
    if (p <= 0.f) return 1.f;
    uint64_t z = idx + ((uint64_t)seed << 32) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // [0, 1)
    return u < p ? 0.f : 1.0f / (1.0f - p);
}

__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, int64_t ld, int R, int C,
                                                            float* __restrict__ out, int Rp) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(int64_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(int64_t)c * Rp + r] = tile[tx][i];
    }
}

// out[c] = sum_r w(r) x[r][c],  w(r) = (wt ? wt[r] : 1) * (sel ? sel[r] == sel_value : 1).  One block per 32 columns (a row
// segment = one 128-B line), 32 row lanes of 32 threads, four independent partial sums per thread so that four rows are in
// flight per lane, then a fixed-order LDS reduction over the row lanes (deterministic: no atomics).
constexpr int COLSUM_THIN_ROWS = 32;
inline int colsum_blocks(int R, int C) { return R <= COLSUM_THIN_ROWS ? (C + 1023) / 1024 : (C + 31) / 32; }
__device__ __forceinline__ void weighted_colsum_block(const float* __restrict__ x, int64_t ldx, const float* __restrict__ wt,
                                                      const int32_t* __restrict__ sel, int sel_value, int R, int C,
                                                      float* __restrict__ out, int block) {
    if (R <= COLSUM_THIN_ROWS) {
        // few rows (the position table's gradient: 5 rows x T * 768 columns): a thread per column, 1024 columns per block.  The wide
        // form below gives row r to lane row r alone and then adds the 32 lane-row sums in order, i.e. the plain sum over r: same bits.
        const int c = block * 1024 + threadIdx.x;
        if (c >= C) return;
        float t = 0.f;
        for (int r = 0; r < R; ++r) {
            float w = wt ? wt[r] : 1.f;
            if (sel && sel[r] != sel_value) w = 0.f;
            t += fmaf(w, x[(int64_t)r * ldx + c], 0.f);
        }
        out[c] = t;
        return;
    }
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = block * 32 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto wgt = [&](int r) -> float {
        float w = wt ? wt[r] : 1.f;
        if (sel && sel[r] != sel_value) w = 0.f;
        return w;
    };
    if (c < C) {
        int r = rl;
        for (; r + 96 < R; r += 128) {
            const float x0 = x[(int64_t)r * ldx + c], x1 = x[(int64_t)(r + 32) * ldx + c];
            const float x2 = x[(int64_t)(r + 64) * ldx + c], x3 = x[(int64_t)(r + 96) * ldx + c];
            s0 = fmaf(wgt(r), x0, s0); s1 = fmaf(wgt(r + 32), x1, s1);
            s2 = fmaf(wgt(r + 64), x2, s2); s3 = fmaf(wgt(r + 96), x3, s3);
        }
        for (; r < R; r += 32) s0 = fmaf(wgt(r), x[(int64_t)r * ldx + c], s0);
    }
    red[rl][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][col];
        out[c] = t;
    }
}

__global__ __launch_bounds__(1024) void weighted_colsum_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ wt,
                                                               const int32_t* __restrict__ sel, int sel_value, int R, int C,
                                                               float* __restrict__ out) {
    weighted_colsum_block(x, ldx, wt, sel, sel_value, R, C, out, blockIdx.x);
}

// Many column sums in one launch (the ~36 bias / LayerNorm / embedding gradients of a training step, each a 5-7 us kernel of a few
// blocks): the item table travels in the kernel arguments, a block finds its item by a scan over the first-block numbers (uniform,
// scalar loads) and then runs exactly the single-matrix block above — the same bits.
struct ColsumGroup { hirest_colsum_item item[HIREST_COLSUM_GROUP_MAX]; int first[HIREST_COLSUM_GROUP_MAX]; int count; };
__global__ __launch_bounds__(1024) void weighted_colsum_grouped_kernel(ColsumGroup g) {
    int i = 0;                                               // first[] ascends: the item is the number of later items starting at or before
#pragma unroll                                               // this block (independent scalar loads; a dependent scan cost 3 us per block)
    for (int j = 1; j < HIREST_COLSUM_GROUP_MAX; ++j) i += (j < g.count && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
    const hirest_colsum_item& it = g.item[i];
    weighted_colsum_block(it.x, it.ldx, it.row_weight, it.row_select, it.select_value, it.R, it.C, it.out, blockIdx.x - g.first[i]);
}

__device__ __forceinline__ float gelu_grad(float x) {      // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
    return 0.5f * erfcf(-x * 0.7071067811865476f) + x * phi;
}

__global__ void act_kernel(const float* __restrict__ pre, float* __restrict__ y, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    y[i] = act == 1 ? 0.5f * x * erfcf(-x * 0.7071067811865476f) : act == 2 ? tanhf(x) : x;
}

__global__ void act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy, float* __restrict__ dx, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    float g = 1.f;
    if (act == 1) g = gelu_grad(x);
    else if (act == 2) { const float t = tanhf(x); g = 1.f - t * t; }
    else if (act == 3) g = 1.f - x * x;                      // `pre` holds y = tanh(pre): the forward kernel kept only that
    dx[i] = dy[i] * g;
}

__global__ void dropout_kernel(const float* __restrict__ x, const float* __restrict__ resid, float* __restrict__ y, int64_t n,
                               float p, uint32_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (resid ? resid[i] : 0.f) + x[i] * keep_scale(seed, (uint64_t)i, p);
}

// LayerNorm backward, one wave per row: xhat = (x - mean) rstd (biased variance, eps inside the root), g = dy * gamma,
// dx = rstd (g - mean(g) - xhat mean(g xhat));  dyxhat = dy * xhat (column sums of it = dgamma; column sums of dy = dbeta)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                            float* __restrict__ dyxhat, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + (int64_t)row * D;
    const float* dr = dy + (int64_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum_x(s) / D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum_x(q) / D + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mean) * rstd, g = dr[c] * gamma[c];
        a += g; b = fmaf(g, xh, b);
    }
    a = wave_sum_x(a) / D; b = wave_sum_x(b) / D;
    for (int c = lane; c < D; c += 64) {
        const float xh = (xr[c] - mean) * rstd, g = dr[c] * gamma[c];
        dx[(int64_t)row * D + c] = rstd * __builtin_fmaf(-xh, b, g - a);
        dyxhat[(int64_t)row * D + c] = dr[c] * xh;
    }
}

// NE = D / 64 elements per lane held in registers (one burst of loads; the generic form below re-reads the row four times);
// same element -> lane assignment, same order of every sum: same bits.
template <int NE>
__global__ __launch_bounds__(256) void layernorm_bwd_regs_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ gamma, float eps, float* __restrict__ dx,
                                                                 float* __restrict__ dyxhat, int R, int D) {
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < R;
    row = live ? row : R - 1;                                // (no early exit in front of the cross-lane reductions; stores are guarded)
    const float* xr = x + (int64_t)row * D;
    const float* dr = dy + (int64_t)row * D;
    float xv[NE], dv[NE], gv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { xv[i] = xr[lane + 64 * i]; dv[i] = dr[lane + 64 * i]; gv[i] = gamma[lane + 64 * i]; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) s += xv[i];
    const float mean = wave_sum_x(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) { const float d = xv[i] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum_x(q) / D + eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const float xh = (xv[i] - mean) * rstd, g = dv[i] * gv[i];
        a += g; b = fmaf(g, xh, b);
    }
    a = wave_sum_x(a) / D; b = wave_sum_x(b) / D;
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const float xh = (xv[i] - mean) * rstd, g = dv[i] * gv[i];
        dx[(int64_t)row * D + lane + 64 * i] = rstd * __builtin_fmaf(-xh, b, g - a);
        dyxhat[(int64_t)row * D + lane + 64 * i] = dv[i] * xh;
    }
}

// ---- attention that keeps its probabilities (training).  General form: q [B*Tq, ldq], k / v [B*Tk, ldkv] (head h at
// columns 64 h ..), optional additive mask [B, Tq, Tk] (the decoder's -10000 on future / padded keys, module_decoder.py:394-397)
// on top of the uniform add_const (the -10000 the all-zeros encoder masks turn into).
// forward, one wave per (b, h, i): P[b,h,i,:] = softmax_j(fl(fl(q_i.k_j * scale) + m_ij)), ctx_i = sum_j drop(P_ij) v_j
struct AttnT {
    const float* q; int64_t ldq;
    const float* k; const float* v; int64_t ldkv;
    const float* mask;
    int B, Tq, Tk, H;
    float scale, addc, drop;
    uint32_t seed;
};

__global__ __launch_bounds__(64) void attention_train_fwd_kernel(AttnT a, float* __restrict__ P, float* __restrict__ ctx, int64_t ldctx) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;                       // (b*H + h)*Tq + i
    const int i = row % a.Tq, bh = row / a.Tq, h = bh % a.H, b = bh / a.H;
    const float* qi = a.q + (int64_t)(b * a.Tq + i) * a.ldq + h * 64;
    const float* mrow = a.mask ? a.mask + (int64_t)(b * a.Tq + i) * a.Tk : nullptr;
    float* prow = P + row * a.Tk;
    __shared__ float qs[64];
    qs[lane] = qi[lane];
    __syncthreads();
    float mx = -3.0e38f;
    for (int j = lane; j < a.Tk; j += 64) {
        const float* kj = a.k + (int64_t)(b * a.Tk + j) * a.ldkv + h * 64;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) s = fmaf(qs[d], kj[d], s);
        s = s * a.scale + (a.addc + (mrow ? mrow[j] : 0.f));
        prow[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max_x(mx);
    float sum = 0.f;
    for (int j = lane; j < a.Tk; j += 64) { const float e = __expf(prow[j] - mx); prow[j] = e; sum += e; }
    sum = wave_sum_x(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < a.Tk; j += 64) prow[j] *= inv;
    __syncthreads();                                      // one wave: orders the row's global writes before the re-reads below
    float acc = 0.f;                                      // lane = output dim d
    for (int j = 0; j < a.Tk; ++j) {
        const float p = prow[j] * keep_scale(a.seed, (uint64_t)row * a.Tk + j, a.drop);
        acc = fmaf(p, a.v[(int64_t)(b * a.Tk + j) * a.ldkv + h * 64 + lane], acc);
    }
    ctx[(int64_t)(b * a.Tq + i) * ldctx + h * 64 + lane] = acc;
}

// backward A, one wave per (b, h, i): dP~_j = dctx_i . v_j, dP_j = dP~_j keep_j, dS_j = P_j (dP_j - sum_j P_j dP_j);
// writes dS[b,h,i,:] and dq_i = scale sum_j dS_j k_j
__global__ __launch_bounds__(64) void attention_train_bwd_q_kernel(AttnT a, const float* __restrict__ P, const float* __restrict__ dctx,
                                                                   int64_t ldctx, float* __restrict__ dS, float* __restrict__ dq,
                                                                   int64_t lddq) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int i = row % a.Tq, bh = row / a.Tq, h = bh % a.H, b = bh / a.H;
    __shared__ float dc[64];
    dc[lane] = dctx[(int64_t)(b * a.Tq + i) * ldctx + h * 64 + lane];
    __syncthreads();
    const float* prow = P + row * a.Tk;
    float* dsrow = dS + row * a.Tk;
    float delta = 0.f;
    for (int j = lane; j < a.Tk; j += 64) {
        const float* vj = a.v + (int64_t)(b * a.Tk + j) * a.ldkv + h * 64;
        float dp = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) dp = fmaf(dc[d], vj[d], dp);
        dp *= keep_scale(a.seed, (uint64_t)row * a.Tk + j, a.drop);
        dsrow[j] = dp;
        delta = fmaf(prow[j], dp, delta);
    }
    delta = wave_sum_x(delta);
    for (int j = lane; j < a.Tk; j += 64) dsrow[j] = prow[j] * (dsrow[j] - delta);
    __syncthreads();
    float acc = 0.f;
    for (int j = 0; j < a.Tk; ++j) acc = fmaf(dsrow[j], a.k[(int64_t)(b * a.Tk + j) * a.ldkv + h * 64 + lane], acc);
    dq[(int64_t)(b * a.Tq + i) * lddq + h * 64 + lane] = acc * a.scale;
}

// backward B, one wave per (b, h, j): dk_j = scale sum_i dS_ij q_i,  dv_j = sum_i drop(P_ij) dctx_i   (lane = dim)
__global__ __launch_bounds__(64) void attention_train_bwd_kv_kernel(AttnT a, const float* __restrict__ P, const float* __restrict__ dctx,
                                                                    int64_t ldctx, const float* __restrict__ dS, float* __restrict__ dk,
                                                                    float* __restrict__ dv, int64_t lddkv) {
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;                       // (b*H + h)*Tk + j
    const int j = row % a.Tk, bh = row / a.Tk, h = bh % a.H, b = bh / a.H;
    float gk = 0.f, gv = 0.f;
    for (int i = 0; i < a.Tq; ++i) {
        const int64_t pi = ((int64_t)bh * a.Tq + i) * a.Tk + j;
        const float ds = dS[pi];
        const float p = P[pi] * keep_scale(a.seed, (uint64_t)pi, a.drop);
        gk = fmaf(ds, a.q[(int64_t)(b * a.Tq + i) * a.ldq + h * 64 + lane], gk);
        gv = fmaf(p, dctx[(int64_t)(b * a.Tq + i) * ldctx + h * 64 + lane], gv);
    }
    dk[(int64_t)(b * a.Tk + j) * lddkv + h * 64 + lane] = gk * a.scale;
    dv[(int64_t)(b * a.Tk + j) * lddkv + h * 64 + lane] = gv;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same attention as tiled matrix products (round 3).  One wave per score row (above) ran the five products of a layer —
// S = Q K^T, ctx = drop(P) V, dP = dctx V^T, dq = dS K, dk = dS^T Q, dv = drop(P)^T dctx — as 64-deep scalar dot products:
// 1.2 ms per layer at B = 5, T = 300 (2.8 TFLOP/s).  Here each product is a batched 64 x 64-tile GEMM on
// v_mfma_f32_32x32x2_f32 (one batch entry per (b, h): grid.z), operands addressed through element strides so that transposed
// and head-sliced views need no copy, the dropout mask regenerated inside the operand load or the epilogue, and the row-wise
// steps (softmax; dS = P (dP - sum P dP)) are one wave per row over rows that are already in memory.
//   C[z][m][n] = epi( sum_k A[z][m][k] B[z][n][k] ),   z = (b1, b2): pointer offsets b1 * s?b1 + b2 * s?b2
// Tile / MFMA layout as hirest_gemm_f32's 64x64 kernel (joint.hip): 4 waves (2 x 2) of one 32x32 tile, K staged 32 deep through
// LDS with the (2 (k & 15) + (k >> 4)) slab image; summation in k order over slabs (no K-quarter split: these are their own ops).
struct BGemm {
    const float* A; int64_t sam, sak, sab1, sab2;
    const float* B; int64_t sbn, sbk, sbb1, sbb2;
    float* C; int64_t scm, scb1, scb2;          // C[m][n] at scm * m + n
    int M, N, K, nb2;
    float alpha;                                // epilogue: acc * alpha
    float addc; const float* mask; int64_t smb1;   // scores: + (addc + mask[b1][m][n]) (mask NULL: + addc), mask row stride N
    int drop_a;                                 // 1: A[m][k] is P[m][k] and gets its dropout keep factor; 2: A[m][k] = P[k][m] (transposed view)
    int drop_c;                                 // 1: epilogue multiplies by the keep factor of P[m][n]
    float drop; uint32_t seed; int Tq, Tk;      // mask geometry: element (i, j) of batch entry z has index (z * Tq + i) * Tk + j
    int a_vec, b_vec;                           // the operand's strides and base allow aligned 16-byte loads along its contiguous dimension
};
inline int vec_ok(const float* p, int64_t s0, int64_t s1, int64_t sb1, int64_t sb2) {   // one stride is 1, the others multiples of 4
    const int64_t big = s0 == 1 ? s1 : s0;
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (s0 == 1 || s1 == 1) && big % 4 == 0 && sb1 % 4 == 0 && sb2 % 4 == 0) ? 1 : 0;
}
inline void set_vec(BGemm& g) { g.a_vec = vec_ok(g.A, g.sam, g.sak, g.sab1, g.sab2); g.b_vec = vec_ok(g.B, g.sbn, g.sbk, g.sbb1, g.sbb2); }
constexpr int BK_ = 32, BLD = BK_ + 1;

__global__ __launch_bounds__(256) void gemm_f32_batched_kernel(BGemm p) {
    __shared__ float As[64 * BLD];
    __shared__ float Bs[64 * BLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M0 = blockIdx.y * 64, N0 = blockIdx.x * 64, z = blockIdx.z;
    const int b1 = z / p.nb2, b2 = z - b1 * p.nb2;
    const float* Ab = p.A + b1 * p.sab1 + b2 * p.sab2;
    const float* Bb = p.B + b1 * p.sbb1 + b2 * p.sbb2;
    // staging: 64 rows x 32 k per operand and slab, 8 elements per thread.  The thread -> element map follows the operand's
    // contiguous dimension so that a wave's loads are whole lines either way:
    //   k contiguous (s?k == 1): thread -> row tid / 4, k = 8 (tid % 4) .. + 7
    //   rows contiguous        : thread -> k = tid / 8 (0..31), rows 8 (tid % 8) .. + 7
    const bool a_kc = p.sak == 1, b_kc = p.sbk == 1;
    const int a_r0 = a_kc ? tid >> 2 : (tid & 7) * 8, a_k0 = a_kc ? (tid & 3) * 8 : tid >> 3;
    const int b_r0 = b_kc ? tid >> 2 : (tid & 7) * 8, b_k0 = b_kc ? (tid & 3) * 8 : tid >> 3;
    float av[8], bv[8];
    // 16-byte loads where the 8 elements of a thread are contiguous, in range and aligned (p.a_vec / p.b_vec: strides and base
    // are multiples of 4 floats); element-wise otherwise (edges, odd strides)
    auto fetch_op = [&](const float* base, int64_t s_r, int64_t s_k, bool kc, int r0, int kk0, int R0, int Rn, int k0, bool vec, int dropm,
                        float (&o)[8]) {
        const int r = R0 + r0, k = k0 + kk0;
        const bool full = kc ? (r < Rn && k + 8 <= p.K) : (r + 8 <= Rn && k < p.K);
        if (vec && full && !dropm) {
            const float* ptr = base + (int64_t)r * s_r + (int64_t)k * s_k;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(ptr), x1 = *reinterpret_cast<const f32x4*>(ptr + 4);
            o[0] = x0[0]; o[1] = x0[1]; o[2] = x0[2]; o[3] = x0[3]; o[4] = x1[0]; o[5] = x1[1]; o[6] = x1[2]; o[7] = x1[3];
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int rr = r + (kc ? 0 : e), kk = k + (kc ? e : 0);
            const bool in = rr < Rn && kk < p.K;
            float v = base[(int64_t)(in ? rr : 0) * s_r + (int64_t)(in ? kk : 0) * s_k];
            if (dropm) {
                const int pi = dropm == 1 ? rr : kk, pj = dropm == 1 ? kk : rr;
                v *= keep_scale(p.seed, ((uint64_t)z * p.Tq + pi) * p.Tk + pj, p.drop);
            }
            o[e] = in ? v : 0.f;
        }
    };
    auto fetch = [&](int k0) {
        fetch_op(Ab, p.sam, p.sak, a_kc, a_r0, a_k0, M0, p.M, k0, p.a_vec != 0, p.drop_a, av);
        fetch_op(Bb, p.sbn, p.sbk, b_kc, b_r0, b_k0, N0, p.N, k0, p.b_vec != 0, 0, bv);
    };
    const int arow = (wm * 32 + (lane & 31)) * BLD + (lane >> 5);
    const int brow = (wn * 32 + (lane & 31)) * BLD + (lane >> 5);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nslab = (p.K + BK_ - 1) / BK_;
    fetch(0);
    for (int sl = 0; sl < nslab; ++sl) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ar = a_r0 + (a_kc ? 0 : e), ak = a_k0 + (a_kc ? e : 0);
            As[ar * BLD + 2 * (ak & 15) + (ak >> 4)] = av[e];
            const int br = b_r0 + (b_kc ? 0 : e), bk = b_k0 + (b_kc ? e : 0);
            Bs[br * BLD + 2 * (bk & 15) + (bk >> 4)] = bv[e];
        }
        __syncthreads();
        if (sl + 1 < nslab) fetch((sl + 1) * BK_);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Bs[brow + 2 * j], As[arow + 2 * j], acc, 0, 0, 0);
    }
    const int m = M0 + wm * 32 + (lane & 31);
    if (m >= p.M) return;
    float* crow = p.C + b1 * p.scb1 + b2 * p.scb2 + (int64_t)m * p.scm;
    const float* mrow = p.mask ? p.mask + b1 * p.smb1 + (int64_t)m * p.N : nullptr;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = N0 + wn * 32 + 8 * g + 4 * (lane >> 5) + e;
            if (n >= p.N) continue;
            float v = acc[4 * g + e] * p.alpha;
            if (p.mask || p.addc != 0.f) v = v + (p.addc + (mrow ? mrow[n] : 0.f));      // fl(fl(q.k * scale) + m), as the reference adds it
            if (p.drop_c) v *= keep_scale(p.seed, ((uint64_t)z * p.Tq + m) * p.Tk + n, p.drop);
            crow[n] = v;
        }
}

// P[row] = softmax(P[row]) in place (row = (b, h, i), Tk scores), one wave per row
__global__ __launch_bounds__(256) void attn_softmax_rows_kernel(float* __restrict__ P, int64_t rows, int Tk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* pr = P + row * Tk;
    float mx = -3.0e38f;
    for (int j = lane; j < Tk; j += 64) mx = fmaxf(mx, pr[j]);
    mx = wave_max_x(mx);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) { const float e = __expf(pr[j] - mx); pr[j] = e; sum += e; }
    sum = wave_sum_x(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tk; j += 64) pr[j] *= inv;
}

// dS[row] = P[row] * (dP[row] - sum_j P dP) in place over dS (which holds dP), one wave per row
__global__ __launch_bounds__(256) void attn_ds_rows_kernel(const float* __restrict__ P, float* __restrict__ dS, int64_t rows, int Tk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = P + row * Tk;
    float* dr = dS + row * Tk;
    float delta = 0.f;
    for (int j = lane; j < Tk; j += 64) delta = fmaf(pr[j], dr[j], delta);
    delta = wave_sum_x(delta);
    for (int j = lane; j < Tk; j += 64) dr[j] = pr[j] * (dr[j] - delta);
}

std::atomic<int> g_attn_train_tiled{1};     // 0: the one-wave-per-row kernels (A/B, tests)

// x[r] = table[ids[r]] + pos[r % T]   (DecoderEmbeddings, module_decoder.py:309-321) and its scatter-add backward
__global__ void embedding_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table, const float* __restrict__ pos,
                                     float* __restrict__ out, int64_t rows, int T, int D, const int32_t* __restrict__ pos_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int c = i - r * D;
    const int64_t pr = pos_ids ? pos_ids[r] : r % T;     // explicit positions: packed ragged sequences
    out[i] = table[(int64_t)ids[r] * D + c] + pos[pr * D + c];
}
__global__ void embedding_bwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dx, float* __restrict__ dtable,
                                     int64_t rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int c = i - r * D;
    atomicAdd(dtable + (int64_t)ids[r] * D + c, dx[i]);
}

// CrossEntropyLoss(ignore_index = -1) over vocabulary rows (modeling.py:140, modeling.py:519): one workgroup per row;
// *loss += weight * (lse - logit[target]) / n_valid, dlogits = weight (softmax - onehot) / n_valid, rows with target < 0: zero
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ target,
                                                      int V, float weight, float inv_valid, float* __restrict__ loss,
                                                      float* __restrict__ dlogits) {
    __shared__ float red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* lr = logits + (int64_t)r * ld;
    float* dr = dlogits + (int64_t)r * ld;
    const int tg = target[r];
    if (tg < 0) { for (int c = tid; c < V; c += 256) dr[c] = 0.f; return; }
    float mx = -3.0e38f;
    for (int c = tid; c < V; c += 256) mx = fmaxf(mx, lr[c]);
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float s = 0.f;
    for (int c = tid; c < V; c += 256) s += __expf(lr[c] - mx);
    red[tid] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float lse = mx + __logf(red[0]);
    for (int c = tid; c < V; c += 256) dr[c] = weight * inv_valid * (__expf(lr[c] - lse) - (c == tg ? 1.f : 0.f));
    if (tid == 0) atomicAdd(loss, weight * inv_valid * (lse - lr[tg]));
}

// masked BCE-with-logits of one head against a one-hot target (modeling.py:249-263):
//   loss += weight * sum_{b,t} mask[b,t] * bce(logit[b,t], t == target[b]) / max(sum mask, 1),  dlogits = d loss / d logit
__global__ __launch_bounds__(256) void bce_masked_kernel(const float* __restrict__ logits, const int32_t* __restrict__ target,
                                                         const int32_t* __restrict__ mask, int B, int T, float weight,
                                                         float* __restrict__ loss, float* __restrict__ dlogits) {
    __shared__ float red[256];
    __shared__ float cnt[256];
    const int n = B * T;
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / T, t = i - b * T;
        const float m = (float)mask[i], x = logits[i], y = (t == target[b]) ? 1.f : 0.f;
        // max(x, 0) - x y + log(1 + exp(-|x|)): torch's stable form
        const float l = fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
        s = fmaf(m, l, s);
        c += m;
    }
    red[threadIdx.x] = s; cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; cnt[threadIdx.x] += cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    const float denom = fmaxf(cnt[0], 1.f);
    if (threadIdx.x == 0) *loss += weight * red[0] / denom;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / T, t = i - b * T;
        const float x = logits[i], y = (t == target[b]) ? 1.f : 0.f;
        dlogits[i] = weight * (float)mask[i] * (1.0f / (1.0f + __expf(-x)) - y) / denom;
    }
}

// fusion backward (modeling.py:163 feats = v * tn[:,None]):  dv = dbase * tn[b],  dtn[b] = sum_t dbase[b,t] * v[b,t]
__global__ __launch_bounds__(256) void joint_base_bwd_kernel(const float* __restrict__ dbase, const float* __restrict__ v,
                                                             const float* __restrict__ tn, float* __restrict__ dv,
                                                             float* __restrict__ dtn, int B, int T, int E) {
    const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float t = tn[(int64_t)b * E + e];
    float acc = 0.f;
    // the sum over t stays one fma chain in t order (same bits), but its loads go out 48 rows at a time: one row per
    // iteration was 300 dependent memory round trips in ten blocks (119 us, 3 % of the training step; 16 rows: 44 us)
    constexpr int U = 48;
    for (int t0 = 0; t0 < T; t0 += U) {
        float d[U], x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tt = t0 + u < T ? t0 + u : T - 1;
            const int64_t i = ((int64_t)b * T + tt) * E + e;
            d[u] = dbase[i]; x[u] = v[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (t0 + u < T) {
                dv[((int64_t)b * T + t0 + u) * E + e] = d[u] * t;
                acc = fmaf(d[u], x[u], acc);
            }
    }
    dtn[(int64_t)b * E + e] = acc;
}

// tn = t / |t|  ->  dt = (dtn - tn (tn . dtn)) / |t|     (one wave per row)
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ t, const float* __restrict__ dtn,
                                                        float* __restrict__ dt, int E) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* tr = t + (int64_t)b * E;
    const float* dr = dtn + (int64_t)b * E;
    float q = 0.f, d = 0.f;
    for (int e = lane; e < E; e += 64) { q = fmaf(tr[e], tr[e], q); d = fmaf(tr[e], dr[e], d); }
    q = wave_sum_x(q); d = wave_sum_x(d);
    const float inv = 1.0f / sqrtf(q);
    for (int e = lane; e < E; e += 64) dt[(int64_t)b * E + e] = (dr[e] - tr[e] * inv * (d * inv)) * inv;
}

// dfeats[r][c] = sum_h dl[h*rows + r] * w_h[c]   (heads are Linear(D, 1): modeling.py:80-99)
__global__ void heads_bwd_kernel(const float* __restrict__ dl, int64_t rows, int D, int nheads, const float* __restrict__ w0,
                                 const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ dfeats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int64_t r = i / D;
    const int c = i - r * D;
    float acc = dl[r] * w0[c];
    if (nheads > 1) acc = fmaf(dl[rows + r], w1[c], acc);
    if (nheads > 2) acc = fmaf(dl[2 * rows + r], w2[c], acc);
    dfeats[i] = acc;
}

// cross-entropy over frames of masked logits (modeling.py:343-344: logits[mask == 0] = -finfo.max, then F.cross_entropy with
// mean reduction): one wave per sample.  *loss += weight * (lse - logit[target]) / B;  dlogits = weight (softmax - onehot) / B on
// the frames of the moment, 0 outside (the in-place fill cuts their gradient).
__global__ __launch_bounds__(64) void ce_masked_kernel(const float* __restrict__ logits, const int32_t* __restrict__ mask,
                                                       const int32_t* __restrict__ target, int B, int T, float weight,
                                                       float* __restrict__ loss, float* __restrict__ dlogits) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* lr = logits + (int64_t)b * T;
    const int32_t* mr = mask + (int64_t)b * T;
    const float NEG = -3.4028234663852886e38f;
    float mx = NEG;
    for (int t = lane; t < T; t += 64) mx = fmaxf(mx, mr[t] ? lr[t] : NEG);
    mx = wave_max_x(mx);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) sum += __expf((mr[t] ? lr[t] : NEG) - mx);
    sum = wave_sum_x(sum);
    const float lse = mx + __logf(sum);
    const int tg = target[b];
    for (int t = lane; t < T; t += 64) {
        const float x = mr[t] ? lr[t] : NEG;
        const float p = __expf(x - lse);
        dlogits[(int64_t)b * T + t] = mr[t] ? weight * (p - (t == tg ? 1.f : 0.f)) / B : 0.f;
    }
    if (lane == 0) atomicAdd(loss, weight * (lse - (mr[tg] ? lr[tg] : NEG)) / B);
}

inline dim3 grid1(int64_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

#define S_(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int hirest_transpose_pad_f32(const float* in, int64_t ld_in, int32_t R, int32_t C, float* out, int32_t Rp, void* stream) {
    if (!in || !out || R <= 0 || C <= 0 || Rp < R) return HIREST_E_BADARG;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((C + 31) / 32, (Rp + 31) / 32), dim3(256), 0, S_(stream), in, ld_in, R, C, out, Rp);
    return hirest_launch_status();
}

extern "C" int hirest_weighted_colsum_f32(const float* x, int64_t ldx, const float* row_weight, const int32_t* row_select,
                                          int32_t select_value, int32_t R, int32_t C, float* out, void* stream) {
    if (!x || !out || R <= 0 || C <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(weighted_colsum_kernel, dim3(colsum_blocks(R, C)), dim3(1024), 0, S_(stream), x, ldx, row_weight, row_select,
                       select_value, R, C, out);
    return hirest_launch_status();
}

extern "C" int hirest_weighted_colsum_grouped_f32(const hirest_colsum_item* items, int32_t count, void* stream) {
    if (!items || count <= 0) return HIREST_E_BADARG;
    for (int i = 0; i < count; ++i)
        if (!items[i].x || !items[i].out || items[i].R <= 0 || items[i].C <= 0) return HIREST_E_BADARG;
    for (int base = 0; base < count; base += HIREST_COLSUM_GROUP_MAX) {
        ColsumGroup g;
        g.count = count - base < HIREST_COLSUM_GROUP_MAX ? count - base : HIREST_COLSUM_GROUP_MAX;
        int blocks = 0;
        for (int i = 0; i < g.count; ++i) { g.item[i] = items[base + i]; g.first[i] = blocks; blocks += colsum_blocks(items[base + i].R, items[base + i].C); }
        hipLaunchKernelGGL(weighted_colsum_grouped_kernel, dim3(blocks), dim3(1024), 0, S_(stream), g);
    }
    return hirest_launch_status();
}

__global__ void scale_by_scalar_kernel(float* __restrict__ x, const float* __restrict__ scalar, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= *scalar;
}

extern "C" int hirest_scale_by_device_scalar_f32(float* x, const float* scalar, int64_t n, void* stream) {
    if (!x || !scalar || n <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(scale_by_scalar_kernel, grid1(n), dim3(256), 0, S_(stream), x, scalar, n);
    return hirest_launch_status();
}

extern "C" int hirest_act_f32(const float* pre, float* y, int64_t n, int32_t act, void* stream) {
    if (!pre || !y || n <= 0 || act < 0 || act > 2) return HIREST_E_BADARG;
    hipLaunchKernelGGL(act_kernel, grid1(n), dim3(256), 0, S_(stream), pre, y, n, act);
    return hirest_launch_status();
}

extern "C" int hirest_act_bwd_f32(const float* pre, const float* dy, float* dx, int64_t n, int32_t act, void* stream) {
    if (!pre || !dy || !dx || n <= 0 || act < 0 || act > 3) return HIREST_E_BADARG;
    hipLaunchKernelGGL(act_bwd_kernel, grid1(n), dim3(256), 0, S_(stream), pre, dy, dx, n, act);
    return hirest_launch_status();
}

extern "C" int hirest_dropout_add_f32(const float* x, const float* resid, float* y, int64_t n, float p, uint32_t seed, void* stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f)) return HIREST_E_BADARG;
    hipLaunchKernelGGL(dropout_kernel, grid1(n), dim3(256), 0, S_(stream), x, resid, y, n, p, seed);
    return hirest_launch_status();
}

extern "C" int hirest_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* dyxhat,
                                        int32_t R, int32_t D, void* stream) {
    if (!x || !dy || !gamma || !dx || !dyxhat || R <= 0 || D <= 0) return HIREST_E_BADARG;
    const dim3 grid((R + 3) / 4), blk(256);
    if (D == 768) hipLaunchKernelGGL(layernorm_bwd_regs_kernel<12>, grid, blk, 0, S_(stream), x, dy, gamma, eps, dx, dyxhat, R, D);
    else if (D == 512) hipLaunchKernelGGL(layernorm_bwd_regs_kernel<8>, grid, blk, 0, S_(stream), x, dy, gamma, eps, dx, dyxhat, R, D);
    else if (D == 384) hipLaunchKernelGGL(layernorm_bwd_regs_kernel<6>, grid, blk, 0, S_(stream), x, dy, gamma, eps, dx, dyxhat, R, D);
    else if (D == 1024) hipLaunchKernelGGL(layernorm_bwd_regs_kernel<16>, grid, blk, 0, S_(stream), x, dy, gamma, eps, dx, dyxhat, R, D);
    else hipLaunchKernelGGL(layernorm_bwd_kernel, grid, blk, 0, S_(stream), x, dy, gamma, eps, dx, dyxhat, R, D);
    return hirest_launch_status();
}

extern "C" int hirest_attention_train_fwd_qkv_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                                  const float* mask_add, float* P, float* ctx, int64_t ldctx, int32_t B, int32_t Tq,
                                                  int32_t Tk, int32_t H, int32_t dh, float scale, float add_const, float drop_p,
                                                  uint32_t seed, void* stream) {
    if (!q || !k || !v || !P || !ctx || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0) return HIREST_E_BADARG;
    if (dh != 64 || !(drop_p >= 0.f && drop_p < 1.f)) return HIREST_E_SHAPE;
    const AttnT a{q, ldq, k, v, ldkv, mask_add, B, Tq, Tk, H, scale, add_const, drop_p, seed};
    if (!g_attn_train_tiled) {
        hipLaunchKernelGGL(attention_train_fwd_kernel, dim3((unsigned)((int64_t)B * H * Tq)), dim3(64), 0, S_(stream), a, P, ctx, ldctx);
        return hirest_launch_status();
    }
    const dim3 blk(256);
    const int64_t TT = (int64_t)Tq * Tk;
    // S = fl(fl(Q K^T * scale) + (add_const + mask)) -> P (raw scores), batch entry z = b * H + h
    BGemm s_{q, ldq, 1, (int64_t)Tq * ldq, 64, k, ldkv, 1, (int64_t)Tk * ldkv, 64, P, Tk, (int64_t)H * TT, TT, Tq, Tk, 64, H, scale, add_const, mask_add, TT,
             0, 0, 0.f, seed, Tq, Tk, 0, 0};
    set_vec(s_);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3((Tk + 63) / 64, (Tq + 63) / 64, B * H), blk, 0, S_(stream), s_);
    hipLaunchKernelGGL(attn_softmax_rows_kernel, dim3((unsigned)(((int64_t)B * H * Tq + 3) / 4)), blk, 0, S_(stream), P, (int64_t)B * H * Tq, Tk);
    // ctx = drop(P) V: A = P [Tq, Tk] (k contiguous), B[n][k] = V[k][n] (rows contiguous)
    BGemm c_{P, Tk, 1, (int64_t)H * TT, TT, v, 1, ldkv, (int64_t)Tk * ldkv, 64, ctx, ldctx, (int64_t)Tq * ldctx, 64, Tq, 64, Tk, H, 1.0f, 0.f, nullptr, 0,
             1, 0, drop_p, seed, Tq, Tk, 0, 0};
    set_vec(c_);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3(1, (Tq + 63) / 64, B * H), blk, 0, S_(stream), c_);
    return hirest_launch_status();
}

extern "C" int hirest_attention_train_bwd_qkv_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                                  const float* P, const float* dctx, int64_t ldctx, float* dS, float* dq, int64_t lddq,
                                                  float* dk, float* dv, int64_t lddkv, int32_t B, int32_t Tq, int32_t Tk, int32_t H,
                                                  int32_t dh, float scale, float drop_p, uint32_t seed, void* stream) {
    if (!q || !k || !v || !P || !dctx || !dS || !dq || !dk || !dv || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0) return HIREST_E_BADARG;
    if (dh != 64 || !(drop_p >= 0.f && drop_p < 1.f)) return HIREST_E_SHAPE;
    const AttnT a{q, ldq, k, v, ldkv, nullptr, B, Tq, Tk, H, scale, 0.f, drop_p, seed};
    if (!g_attn_train_tiled) {
        hipLaunchKernelGGL(attention_train_bwd_q_kernel, dim3((unsigned)((int64_t)B * H * Tq)), dim3(64), 0, S_(stream), a, P, dctx, ldctx, dS, dq, lddq);
        hipLaunchKernelGGL(attention_train_bwd_kv_kernel, dim3((unsigned)((int64_t)B * H * Tk)), dim3(64), 0, S_(stream), a, P, dctx, ldctx, dS, dk, dv,
                           lddkv);
        return hirest_launch_status();
    }
    const dim3 blk(256);
    const int64_t TT = (int64_t)Tq * Tk;
    const int Z = B * H;
    // dP = keep * (dctx V^T) -> dS buffer
    BGemm dp{dctx, ldctx, 1, (int64_t)Tq * ldctx, 64, v, ldkv, 1, (int64_t)Tk * ldkv, 64, dS, Tk, (int64_t)H * TT, TT, Tq, Tk, 64, H, 1.0f, 0.f, nullptr, 0,
             0, 1, drop_p, seed, Tq, Tk, 0, 0};
    set_vec(dp);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3((Tk + 63) / 64, (Tq + 63) / 64, Z), blk, 0, S_(stream), dp);
    hipLaunchKernelGGL(attn_ds_rows_kernel, dim3((unsigned)(((int64_t)Z * Tq + 3) / 4)), blk, 0, S_(stream), P, dS, (int64_t)Z * Tq, Tk);
    // dq = scale * dS K: A = dS [Tq, Tk], B[n][k] = K[k][n]
    BGemm gq{dS, Tk, 1, (int64_t)H * TT, TT, k, 1, ldkv, (int64_t)Tk * ldkv, 64, dq, lddq, (int64_t)Tq * lddq, 64, Tq, 64, Tk, H, scale, 0.f, nullptr, 0,
             0, 0, 0.f, seed, Tq, Tk, 0, 0};
    set_vec(gq);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3(1, (Tq + 63) / 64, Z), blk, 0, S_(stream), gq);
    // dk = scale * dS^T Q: A[m][k] = dS[k][m] (rows contiguous), B[n][k] = Q[k][n]
    BGemm gk{dS, 1, Tk, (int64_t)H * TT, TT, q, 1, ldq, (int64_t)Tq * ldq, 64, dk, lddkv, (int64_t)Tk * lddkv, 64, Tk, 64, Tq, H, scale, 0.f, nullptr, 0,
             0, 0, 0.f, seed, Tq, Tk, 0, 0};
    set_vec(gk);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3(1, (Tk + 63) / 64, Z), blk, 0, S_(stream), gk);
    // dv = drop(P)^T dctx: A[m][k] = keep * P[k][m], B[n][k] = dctx[k][n]
    BGemm gv{P, 1, Tk, (int64_t)H * TT, TT, dctx, 1, ldctx, (int64_t)Tq * ldctx, 64, dv, lddkv, (int64_t)Tk * lddkv, 64, Tk, 64, Tq, H, 1.0f, 0.f, nullptr, 0,
             2, 0, drop_p, seed, Tq, Tk, 0, 0};
    set_vec(gv);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3(1, (Tk + 63) / 64, Z), blk, 0, S_(stream), gv);
    return hirest_launch_status();
}

// C = alpha * A B^T over arbitrary element strides (A[m][k] at sam m + sak k, B[n][k] at sbn n + sbk k, one of each pair = 1):
// the dX = dY W and dW = dY^T X products of the backward pass read their operands in place instead of through transposed copies.
extern "C" int hirest_gemm_f32_strided(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, float* C,
                                       int64_t ldc, int32_t M, int32_t N, int32_t K, float alpha, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return HIREST_E_BADARG;
    if ((sam != 1 && sak != 1) || (sbn != 1 && sbk != 1)) return HIREST_E_SHAPE;
    BGemm g{A, sam, sak, 0, 0, B, sbn, sbk, 0, 0, C, ldc, 0, 0, M, N, K, 1, alpha, 0.f, nullptr, 0, 0, 0, 0.f, 0u, 1, 1, 0, 0};
    set_vec(g);
    hipLaunchKernelGGL(gemm_f32_batched_kernel, dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, S_(stream), g);
    return hirest_launch_status();
}

extern "C" int hirest_attention_train_select(int32_t which) {
    if (which != 0 && which != 1) return HIREST_E_BADARG;
    g_attn_train_tiled = which;
    return 0;
}

// packed self-attention forms (q | k | v in one [B*T, 3*H*64] activation)
extern "C" int hirest_attention_train_fwd_f32(const float* qkv, float* P, float* ctx, int32_t B, int32_t T, int32_t H, int32_t dh,
                                              float scale, float add_const, float drop_p, uint32_t seed, void* stream) {
    if (!qkv) return HIREST_E_BADARG;
    const int64_t D = (int64_t)H * 64;
    return hirest_attention_train_fwd_qkv_f32(qkv, 3 * D, qkv + D, qkv + 2 * D, 3 * D, nullptr, P, ctx, D, B, T, T, H, dh, scale, add_const,
                                              drop_p, seed, stream);
}

extern "C" int hirest_attention_train_bwd_f32(const float* qkv, const float* P, const float* dctx, float* dS, float* dqkv, int32_t B,
                                              int32_t T, int32_t H, int32_t dh, float scale, float drop_p, uint32_t seed, void* stream) {
    if (!qkv || !dqkv) return HIREST_E_BADARG;
    const int64_t D = (int64_t)H * 64;
    return hirest_attention_train_bwd_qkv_f32(qkv, 3 * D, qkv + D, qkv + 2 * D, 3 * D, P, dctx, D, dS, dqkv, 3 * D, dqkv + D, dqkv + 2 * D,
                                              3 * D, B, T, T, H, dh, scale, drop_p, seed, stream);
}

extern "C" int hirest_embedding_fwd_f32(const int32_t* ids, const float* table, const float* pos, float* out, int64_t rows, int32_t T,
                                        int32_t D, void* stream) {
    if (!ids || !table || !pos || !out || rows <= 0 || T <= 0 || D <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(embedding_fwd_kernel, grid1(rows * D), dim3(256), 0, S_(stream), ids, table, pos, out, rows, T, D, nullptr);
    return hirest_launch_status();
}

extern "C" int hirest_embedding_pos_fwd_f32(const int32_t* ids, const int32_t* pos_ids, const float* table, const float* pos, float* out,
                                            int64_t rows, int32_t D, void* stream) {
    if (!ids || !pos_ids || !table || !pos || !out || rows <= 0 || D <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(embedding_fwd_kernel, grid1(rows * D), dim3(256), 0, S_(stream), ids, table, pos, out, rows, 1, D, pos_ids);
    return hirest_launch_status();
}

extern "C" int hirest_embedding_bwd_f32(const int32_t* ids, const float* dx, float* dtable_accum, int64_t rows, int32_t D, void* stream) {
    if (!ids || !dx || !dtable_accum || rows <= 0 || D <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(embedding_bwd_kernel, grid1(rows * D), dim3(256), 0, S_(stream), ids, dx, dtable_accum, rows, D);
    return hirest_launch_status();
}

extern "C" int hirest_ce_rows_f32(const float* logits, int64_t ld, const int32_t* target, int32_t R, int32_t V, float weight,
                                  int32_t n_valid, float* loss_accum, float* dlogits, void* stream) {
    if (!logits || !target || !loss_accum || !dlogits || R <= 0 || V <= 0 || ld < V) return HIREST_E_BADARG;
    hipLaunchKernelGGL(ce_rows_kernel, dim3(R), dim3(256), 0, S_(stream), logits, ld, target, V, weight, 1.0f / (float)(n_valid > 0 ? n_valid : 1),
                       loss_accum, dlogits);
    return hirest_launch_status();
}

extern "C" int hirest_bce_masked_f32(const float* logits, const int32_t* target, const int32_t* mask, int32_t B, int32_t T, float weight,
                                     float* loss_accum, float* dlogits, void* stream) {
    if (!logits || !target || !mask || !loss_accum || !dlogits || B <= 0 || T <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(bce_masked_kernel, dim3(1), dim3(256), 0, S_(stream), logits, target, mask, B, T, weight, loss_accum, dlogits);
    return hirest_launch_status();
}

extern "C" int hirest_joint_base_bwd_f32(const float* dbase, const float* v, const float* tn, float* dv, float* dtn, int32_t B, int32_t T,
                                         int32_t E, void* stream) {
    if (!dbase || !v || !tn || !dv || !dtn || B <= 0 || T <= 0 || E <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(joint_base_bwd_kernel, dim3((E + 255) / 256, B), dim3(256), 0, S_(stream), dbase, v, tn, dv, dtn, B, T, E);
    return hirest_launch_status();
}

extern "C" int hirest_l2norm_bwd_f32(const float* t, const float* dtn, float* dt, int32_t B, int32_t E, void* stream) {
    if (!t || !dtn || !dt || B <= 0 || E <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(64), 0, S_(stream), t, dtn, dt, E);
    return hirest_launch_status();
}

extern "C" int hirest_heads_bwd_f32(const float* dlogits, int64_t rows, int32_t D, int32_t nheads, const float* w0, const float* w1,
                                    const float* w2, float* dfeats, void* stream) {
    if (!dlogits || !w0 || !dfeats || rows <= 0 || D <= 0 || nheads < 1 || nheads > 3) return HIREST_E_BADARG;
    hipLaunchKernelGGL(heads_bwd_kernel, grid1(rows * D), dim3(256), 0, S_(stream), dlogits, rows, D, nheads, w0, w1, w2, dfeats);
    return hirest_launch_status();
}

extern "C" int hirest_ce_masked_f32(const float* logits, const int32_t* mask, const int32_t* target, int32_t B, int32_t T, float weight,
                                    float* loss_accum, float* dlogits, void* stream) {
    if (!logits || !mask || !target || !loss_accum || !dlogits || B <= 0 || T <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(ce_masked_kernel, dim3(B), dim3(64), 0, S_(stream), logits, mask, target, B, T, weight, loss_accum, dlogits);
    return hirest_launch_status();
}
