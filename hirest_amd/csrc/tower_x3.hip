// "bf16x3" vision tower: the reference's fp32 forward (EVA_clip/vit_model.py:326-351, precision='fp32' of eva_clip.py:90) with every
// linear layer's product formed from bf16 hi + lo splits of BOTH fp32 operands on the bf16 matrix pipe,
//     a w  ~=  w_hi a_lo + w_lo a_hi + w_hi a_hi        (hi = bf16(v), lo = bf16(v - hi); the dropped w_lo a_lo is 2^-16 of the product)
// accumulated in fp32: products carry ~16 mantissa bits (fp32 MFMA: 24, the bf16 towers: 8) at three bf16 MFMAs each, i.e. 3/16 of
// the exact-fp32 matrix cost.  Everything that is not a weight GEMM stays what tower_f32.hip runs: fp32 residual stream, fp32
// LayerNorm, the softmax of the attention (whose two products are split-operand products as well: attention_x3.hip), erf-GELU in fp32,
// fp32 patch embedding and head.
//
// Operand format (HIREST_GEMM_X3, include/hirest_hip.h): a [rows, K] fp32 matrix becomes [rows, 2K] bf16 whose 64-column block c holds
// hi(k = 32c .. 32c+31) | lo(same k).  The split is fused into the kernel that produces the operand: LayerNorm (qkv / fc1 input), GELU
// (fc2 input), a plain pass for the attention output; weights are split once per checkpoint by the host side with the same kernel.
//
// Workspace (B frames, M = B*T tokens): x f32 [M, D] residual stream, h f32 [M, D] attention output, big f32 [M, max(3D, Dm, kpad)]
// qkv / pre-activation / patch rows, a2 bf16 [M, 2D] split operand of qkv / proj / fc1, b2 bf16 [M, 2Dm] split operand of fc2.
#include "common.h"
#include <atomic>
#include "profile.h"

namespace {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct RegionsX { size_t x, h, big, a2, b2, total; };
RegionsX plan_x3(int64_t M, int D, int Dm, int wide) {
    RegionsX r; size_t off = 0;
    r.x = off; off += align256((size_t)M * D * 4);
    r.h = off; off += align256((size_t)M * D * 4);
    r.big = off; off += align256((size_t)M * wide * 4);
    r.a2 = off; off += align256((size_t)M * 2 * D * 2);
    r.b2 = off; off += align256((size_t)M * 2 * Dm * 2);
    r.total = off;
    return r;
}
inline int wide_of(int D, int Dm, int kpad) { int w = 3 * D; if (Dm > w) w = Dm; if (kpad > w) w = kpad; return w; }

#define CHECK(expr) do { int _e = (expr); if (_e != 0) return _e; } while (0)

__device__ __forceinline__ void split_store4(const f32x4& y, bf16_t* orow, int col) {   // col % 4 == 0: 4 columns of one 32-block
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)y[e]; lo[e] = (bf16_t)(y[e] - (float)hi[e]); }
    bf16_t* o = orow + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + 32) = lo;
}

// nn.GELU() in fp32, erf form (vit_model.py:49,59): what the reference's fp32 forward evaluates
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// out[r] = split(act(x[r])): ACT 0 none, 1 gelu(erf).  One thread per 8 consecutive columns (two 16-B loads, two 16-B stores).
template <int ACT>
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo,
                                                    int64_t rows, int D) {
    const int per_row = D >> 3;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) * 8;
        const float* xr = x + r * ldx + c;
        f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr));
        f32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4));
        if constexpr (ACT == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = gelu_exact(v0[e]); v1[e] = gelu_exact(v1[e]); }
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (bf16_t)v0[e]; lo[e] = (bf16_t)(v0[e] - (float)hi[e]);
            hi[4 + e] = (bf16_t)v1[e]; lo[4 + e] = (bf16_t)(v1[e] - (float)hi[4 + e]);
        }
        bf16_t* o = out + r * ldo + (c >> 5) * 64 + (c & 31);
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + 32) = lo;
    }
}



// out[c] = split(x[:, c]): the split operand of a product that contracts over the ROWS of a row-major fp32 matrix (dX = dY W: W^T as the GEMM's
// B operand) without a transposed fp32 copy.  One block per 32 rows x 64 columns: coalesced row reads, an LDS transpose, 128-B stores.
__global__ __launch_bounds__(256) void split2_transposed_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo,
                                                               int rows, int cols) {
    __shared__ float tile[32][65];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 64, t = threadIdx.x;
    {
        const int r = t >> 3, c = (t & 7) * 8;
        const float* src = x + (int64_t)(r0 + r) * ldx + c0 + c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c0 + c + 4 * h + 3 < cols) v = *reinterpret_cast<const f32x4*>(src + 4 * h);
            else
                for (int e = 0; e < 4; ++e) if (c0 + c + 4 * h + e < cols) v[e] = src[4 * h + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][c + 4 * h + e] = v[e];
        }
    }
    __syncthreads();
    const int c = t >> 2, part = (t & 3) * 8;
    if (c0 + c >= cols) return;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = tile[part + e][c];
        hi[e] = (bf16_t)v; lo[e] = (bf16_t)(v - (float)hi[e]);
    }
    bf16_t* o = out + (int64_t)(c0 + c) * ldo + (r0 >> 5) * 64 + part;
    *reinterpret_cast<bf16x8*>(o) = hi;
    *reinterpret_cast<bf16x8*>(o + 32) = lo;
}

// Several matrices split in ONE launch (hirest_split2_grouped_bf16): a training step splits the four weights of every encoder block both ways
// (forward: W as the B operand; backward: W^T) — sixteen launches of 5 - 17 us for 2 blocks, against ~25 us of traffic.  One block per 32 x 64
// input tile; the item table travels in the kernel arguments and a block finds its item by comparing against the running tile counts.
struct SplitItems { hirest_split_item it[HIREST_SPLIT_GROUP_MAX]; int first[HIREST_SPLIT_GROUP_MAX + 1]; int count; };
__global__ __launch_bounds__(256) void split2_grouped_kernel(const SplitItems g) {
    __shared__ float tile[32][65];
    int k = 0;
#pragma unroll
    for (int i = 1; i < HIREST_SPLIT_GROUP_MAX; ++i) k += (i < g.count && (int)blockIdx.x >= g.first[i]) ? 1 : 0;
    const hirest_split_item& it = g.it[k];
    const int local = blockIdx.x - g.first[k], tiles_c = (it.cols + 63) / 64;
    const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 64, t = threadIdx.x;
    const int r = t >> 3, c = (t & 7) * 8;
    f32x4 v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (r0 + r < it.rows) {
        const float* src = it.x + (int64_t)(r0 + r) * it.ldx + c0 + c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (c0 + c + 4 * h + 3 < it.cols) v[h] = *reinterpret_cast<const f32x4*>(src + 4 * h);
            else
                for (int e = 0; e < 4; ++e) if (c0 + c + 4 * h + e < it.cols) v[h][e] = src[4 * h + e];
        }
    }
    bf16_t* out = reinterpret_cast<bf16_t*>(it.out);
    if (!it.transposed) {                                                        // (wave-uniform: one item per block)
        if (r0 + r >= it.rows || c0 + c >= it.cols) return;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (bf16_t)v[0][e]; lo[e] = (bf16_t)(v[0][e] - (float)hi[e]);
            hi[4 + e] = (bf16_t)v[1][e]; lo[4 + e] = (bf16_t)(v[1][e] - (float)hi[4 + e]);
        }
        const int col = c0 + c;
        bf16_t* o = out + (int64_t)(r0 + r) * it.ldo + (col >> 5) * 64 + (col & 31);
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + 32) = lo;
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][c + 4 * h + e] = v[h][e];
    __syncthreads();
    const int cc = t >> 2, part = (t & 3) * 8;
    if (c0 + cc >= it.cols) return;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float w = tile[part + e][cc];                                      // rows past it.rows were loaded as zeros
        hi[e] = (bf16_t)w; lo[e] = (bf16_t)(w - (float)hi[e]);
    }
    bf16_t* o = out + (int64_t)(c0 + cc) * it.ldo + (r0 >> 5) * 64 + part;
    *reinterpret_cast<bf16x8*>(o) = hi;
    *reinterpret_cast<bf16x8*>(o + 32) = lo;
}

// LayerNorm (layernorm_rows' arithmetic: ln_wave_stats / ln_apply) with the split as its store: one wave per row
template <int NV>
__global__ __launch_bounds__(256) void layernorm_split2_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, bf16_t* __restrict__ out,
                                                              int64_t ldo, int rows, int D) {
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
        f32x4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nv ? *reinterpret_cast<const f32x4*>(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float mean, rstd;
        ln_wave_stats<NV>(v, nv, D, eps, lane, mean, rstd);
        bf16_t* orow = out + (int64_t)row * ldo;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) split_store4(ln_apply(v[i], mean, rstd, g[i], b[i]), orow, 4 * c);
        }
    }
}

int gemm_x3(const hirest_bf16* A2, int64_t lda, const hirest_bf16* W2, int64_t ldw, const float* bias, float* out, int64_t ldo, int M, int N,
            int K, int epi, void* stream) {
    hirest_gemm_args a;
    a.struct_size = sizeof(a);
    a.A = A2; a.lda = lda; a.W = W2; a.ldw = ldw; a.bias = bias; a.out = out; a.ldo = ldo;
    a.M = M; a.N = N; a.K = 2 * K; a.epilogue = epi; a.pos = nullptr; a.patches_per_frame = 0; a.aux0 = a.aux1 = nullptr;
    a.flags = HIREST_GEMM_X3;
    return hirest_gemm_bf16(&a, stream);
}

}  // namespace

extern "C" int hirest_split2_bf16(const float* x, int64_t ldx, hirest_bf16* out, int64_t ldo, int64_t rows, int32_t D, int32_t act,
                                  void* stream) {
    if (!x || !out || rows <= 0 || D <= 0 || act < 0 || act > 1) return HIREST_E_BADARG;
    if (D % 32 != 0 || ldx % 4 != 0 || ldo % 8 != 0 || ldo < 2 * (int64_t)D) return HIREST_E_SHAPE;
    const int64_t total = rows * (D / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bf16_t* o = reinterpret_cast<bf16_t*>(out);
    if (act == 1) hipLaunchKernelGGL(split2_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, o, ldo, rows, D);
    else hipLaunchKernelGGL(split2_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, o, ldo, rows, D);
    return hirest_launch_status();
}


extern "C" int hirest_split2_transposed_bf16(const float* x, int64_t ldx, hirest_bf16* out, int64_t ldo, int32_t rows, int32_t cols, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return HIREST_E_BADARG;
    if (rows % 32 != 0 || ldx % 4 != 0 || ldo % 8 != 0 || ldo < 2 * (int64_t)rows) return HIREST_E_SHAPE;
    hipLaunchKernelGGL(split2_transposed_kernel, dim3((cols + 63) / 64, rows / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                       reinterpret_cast<bf16_t*>(out), ldo, rows, cols);
    return hirest_launch_status();
}


extern "C" int hirest_split2_grouped_bf16(const hirest_split_item* items, int32_t count, void* stream) {
    if (!items || count <= 0) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < count; base += HIREST_SPLIT_GROUP_MAX) {
        SplitItems g;
        g.count = count - base < HIREST_SPLIT_GROUP_MAX ? count - base : HIREST_SPLIT_GROUP_MAX;
        int total = 0;
        for (int i = 0; i < g.count; ++i) {
            const hirest_split_item& it = items[base + i];
            if (!it.x || !it.out || it.rows <= 0 || it.cols <= 0) return HIREST_E_BADARG;
            const int64_t k = it.transposed ? it.rows : it.cols;                   // the contracted dimension: whole 32-blocks (rows: zero filled)
            if (it.ldx % 4 != 0 || it.ldo % 8 != 0 || (!it.transposed && it.cols % 32 != 0) || it.ldo < 2 * ((k + 31) / 32 * 32)) return HIREST_E_SHAPE;
            g.it[i] = it;
            g.first[i] = total;
            total += ((it.rows + 31) / 32) * ((it.cols + 63) / 64);
        }
        for (int i = g.count; i <= HIREST_SPLIT_GROUP_MAX; ++i) g.first[i] = total;
        hipLaunchKernelGGL(split2_grouped_kernel, dim3(total), dim3(256), 0, s, g);
    }
    return hirest_launch_status();
}


extern "C" int hirest_layernorm_split2(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, hirest_bf16* out,
                                       int64_t ldo, int32_t rows, int32_t D, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0) return HIREST_E_BADARG;
    if (D <= 0 || D % 32 != 0 || D > 2048 || ldx % 4 != 0 || ldo % 8 != 0 || ldo < 2 * (int64_t)D) return HIREST_E_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bf16_t* o = reinterpret_cast<bf16_t*>(out);
    const int nv = (D / 4 + 63) / 64;
    int grid = (rows + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
#define LS_CASE(NVV) case NVV: hipLaunchKernelGGL((layernorm_split2_kernel<NVV>), dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, eps, o, ldo, rows, D); break;
    switch (nv) { LS_CASE(1) LS_CASE(2) LS_CASE(3) LS_CASE(4) LS_CASE(5) LS_CASE(6) LS_CASE(7) LS_CASE(8) default: return HIREST_E_SHAPE; }
#undef LS_CASE
    return hirest_launch_status();
}

static std::atomic<int> g_x3_gelu_pass{0};     // hirest_vision_x3_select_attention bit 1: GELU + split as a separate pass over an fp32 hidden activation (A/B)
static std::atomic<int> g_x3_attention{0};     // 0: split-operand flash attention (attention_x3.hip); 1: the exact-fp32 attention of tower_f32 (A/B, tests)
extern "C" int hirest_vision_x3_select_attention(int32_t which) {
    if (which < 0 || which > 3) return HIREST_E_BADARG;
    g_x3_attention = which & 1;
    g_x3_gelu_pass = (which >> 1) & 1;
    return 0;
}

extern "C" size_t hirest_vision_workspace_bytes_x3(const hirest_vision_tower_x3* t, int32_t B) {
    if (!t || !t->base || B <= 0) return 0;
    const hirest_vision_tower_f32* f = t->base;
    const int T = (f->image_size / f->patch) * (f->image_size / f->patch) + 1;
    return plan_x3((int64_t)B * T, f->width, f->mlp_dim, wide_of(f->width, f->mlp_dim, f->kpad)).total;
}

extern "C" int hirest_vision_forward_x3(const hirest_vision_tower_x3* t, const void* frames, int32_t in_dtype, int32_t B, float* out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!t || !t->base || !t->blocks || !frames || !out || !workspace || B <= 0 || !t->base->blocks) return HIREST_E_BADARG;
    const hirest_vision_tower_f32* f = t->base;
    if (f->width != f->heads * f->head_dim || f->image_size % f->patch != 0 || f->width % 32 != 0 || f->mlp_dim % 32 != 0 || f->width > 2048)
        return HIREST_E_SHAPE;
    if (f->act != 0 || f->ln_pre_g || f->out_all_tokens) return HIREST_E_SHAPE;     // the EVA tower (GELU, no ln_pre, CLS head) only
    const int G = f->image_size / f->patch, T = G * G + 1, D = f->width, Dm = f->mlp_dim;
    const int64_t M64 = (int64_t)B * T;
    if (M64 > 0x7fffffff) return HIREST_E_SHAPE;
    const int M = (int)M64;
    const RegionsX r = plan_x3(M64, D, Dm, wide_of(D, Dm, f->kpad));
    if (workspace_bytes < r.total) return HIREST_E_WORKSPACE;
    char* ws = reinterpret_cast<char*>(workspace);
    float* x = reinterpret_cast<float*>(ws + r.x);
    float* h = reinterpret_cast<float*>(ws + r.h);
    float* big = reinterpret_cast<float*>(ws + r.big);
    hirest_bf16* a2 = reinterpret_cast<hirest_bf16*>(ws + r.a2);
    hirest_bf16* b2 = reinterpret_cast<hirest_bf16*>(ws + r.b2);
    // patch embedding + cls + pos: tower_f32.hip's own front end (exact fp32; 0.08 % of the tower's products)
    CHECK(hirest_vision_embed_f32(f, frames, in_dtype, B, x, big, stream));
    const float scale = 1.0f / sqrtf((float)f->head_dim);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int l = 0; l < f->layers; ++l) {
        const hirest_block_weights_f32& w = f->blocks[l];
        const hirest_block_weights_x3& w2 = t->blocks[l];
        // (profile records of the non-GEMM kernels: kind LAYERNORM tags 10 LayerNorm + split, 11 split, 12 GELU + split; kind ATTENTION tag 2)
        { HirestProfScope pr(HIREST_PROF_LAYERNORM, 10, M, D, 0, s);
          CHECK(hirest_layernorm_split2(x, D, w.ln1_g, w.ln1_b, f->ln_eps, a2, 2 * D, M, D, stream)); }
        CHECK(gemm_x3(a2, 2 * D, w2.qkv_w2, 2 * D, w.qkv_b, big, 3 * D, M, 3 * D, D, HIREST_EPI_BIAS_F32, stream));
        { HirestProfScope pr(HIREST_PROF_ATTENTION, 2, (int64_t)B * f->heads, T, f->head_dim, s);
          if (g_x3_attention == 0)        // writes proj's split operand itself
              CHECK(hirest_attention_x3_qkv_split2(big, 3 * (int64_t)D, big + D, big + 2 * D, 3 * (int64_t)D, a2, B, T, T, f->heads, f->head_dim, scale,
                                                   stream));
          else
              CHECK(hirest_attention_f32_qkv(big, 3 * (int64_t)D, big + D, big + 2 * D, 3 * (int64_t)D, h, B, T, T, f->heads, f->head_dim, scale, 0.f,
                                             0.f, stream)); }
        if (g_x3_attention != 0) {
            HirestProfScope pr(HIREST_PROF_LAYERNORM, 11, M, D, 0, s);
            CHECK(hirest_split2_bf16(h, D, a2, 2 * D, M, D, 0, stream));
        }
        CHECK(gemm_x3(a2, 2 * D, w2.proj_w2, 2 * D, w.proj_b, x, D, M, D, D, HIREST_EPI_BIAS_RESID_F32, stream));
        { HirestProfScope pr(HIREST_PROF_LAYERNORM, 10, M, D, 0, s);
          CHECK(hirest_layernorm_split2(x, D, w.ln2_g, w.ln2_b, f->ln_eps, a2, 2 * D, M, D, stream)); }
        if (g_x3_gelu_pass == 0) {                              // GELU + split in fc1's epilogue: the hidden activation never exists in fp32
            CHECK(gemm_x3(a2, 2 * D, w2.fc1_w2, 2 * D, w.fc1_b, reinterpret_cast<float*>(b2), 2 * Dm, M, Dm, D, HIREST_EPI_BIAS_GELU_SPLIT2, stream));
        } else {
            CHECK(gemm_x3(a2, 2 * D, w2.fc1_w2, 2 * D, w.fc1_b, big, Dm, M, Dm, D, HIREST_EPI_BIAS_F32, stream));
            HirestProfScope pr(HIREST_PROF_LAYERNORM, 12, M, Dm, 0, s);
            CHECK(hirest_split2_bf16(big, Dm, b2, 2 * Dm, M, Dm, 1, stream));
        }
        CHECK(gemm_x3(b2, 2 * Dm, w2.fc2_w2, 2 * Dm, w.fc2_b, x, D, M, D, Dm, HIREST_EPI_BIAS_RESID_F32, stream));
    }
    CHECK(hirest_layernorm(x, (int64_t)T * D, nullptr, f->norm_g, f->norm_b, f->ln_eps, h, D, 1, B, D, stream));
    CHECK(hirest_gemm_f32(h, D, f->head_w, D, f->head_b, nullptr, 0, nullptr, 0, out, f->embed_dim, B, f->embed_dim, D, 0, stream));
    return 0;
}
