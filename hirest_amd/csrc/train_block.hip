// One post-LN block of the clip4caption VisualModel in train mode, forward and backward, as ONE host call per direction
// (include/hirest_hip.h: hirest_train_block; module_visual.py:132-264 under modeling.py:196-211 / run.py:238-295).
//
// No kernel lives here: the file issues the library's own entry points (hirest_gemm_f32_ws / _layouts, hirest_attention_train_*,
// hirest_dropout_add_f32, hirest_layernorm, hirest_act_*, hirest_layernorm_bwd_f32) in the order hirest_amd/train.py issued them one
// by one — same operands, same order, same bits — because at B = 5, T = 300 the training step was paced by the host: ~150 launches
// per step at 15-20 us each through the host language against 2.8 ms of kernels (profiles/r06/train_step_context.txt,
// tools/r06_train_graph.py).  From here a launch costs the ~3 us of hipLaunchKernel.
//
// Weight-gradient products (dW = dY^T X) go to the caller's side stream behind an event recorded when their dY is complete, as
// train.py's _K.grad_weight did; the column sums (bias / LayerNorm gradients) are appended to the caller's item table and run as
// one grouped launch after the whole backward.
#include "common.h"
#include <math.h>
#include <mutex>

namespace {

#define CHECK(expr) do { int _e = (expr); if (_e != 0) return _e; } while (0)
inline size_t al(size_t floats) { return (floats + 63) & ~(size_t)63; }           // 256-B aligned slices of the scratch

struct Shape {
    int64_t R, W, M3, mlp, PT;                                                   // rows, width, 3 width, mlp, B H T T
    explicit Shape(const hirest_train_block* b)
        : R((int64_t)b->B * b->T), W(b->width), M3(3 * (int64_t)b->width), mlp(b->mlp), PT((int64_t)b->B * b->heads * b->T * b->T) {}
};

inline bool block_ok(const hirest_train_block* b) {
    return b && b->struct_size == sizeof(*b) && b->B > 0 && b->T > 0 && b->heads > 0 && b->width > 0 && b->width % b->heads == 0 &&
           b->width % 16 == 0 && b->mlp > 0 && b->mlp % 16 == 0 && (b->precision == 0 || b->precision == 1);
}

// the fp32 GEMM with its split form when the caller's scratch holds it (same bits either way)
int gemm(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, float* out, int M, int N, int K, void* ws, size_t wsb,
         hipStream_t s) {
    const size_t need = hirest_gemm_f32_workspace_bytes(M, N, K);
    const bool use = need && ws && need <= wsb;
    return hirest_gemm_f32_ws(A, lda, Wt, ldw, bias, nullptr, N, nullptr, 0, out, N, M, N, K, 0, use ? ws : nullptr, use ? wsb : 0, s);
}
int gemm_layouts(const float* A, int64_t lda, int akm, const float* Wt, int64_t ldw, int wkm, const float* resid, int64_t ldr, float* out, int M, int N,
                 int K, void* ws, size_t wsb, hipStream_t s) {
    const size_t need = hirest_gemm_f32_layouts_workspace_bytes(M, N, K);
    const bool use = need && ws && need <= wsb;
    return hirest_gemm_f32_layouts(A, lda, akm, Wt, ldw, wkm, nullptr, resid, resid ? ldr : 0, out, N, M, N, K, 0, use ? ws : nullptr, use ? wsb : 0, s);
}

// "dY is complete on the main stream" markers for the side stream: a small ring of events (a wait captures the record that precedes it,
// so reusing an event later does not disturb the waits already enqueued)
hipEvent_t next_event() {
    constexpr int MAXDEV = 64, RING = 64;
    static std::mutex mu;
    static hipEvent_t ring[MAXDEV][RING];
    static int n[MAXDEV] = {}, at[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;      // events belong to the device they were created on
    std::lock_guard<std::mutex> lock(mu);
    if (n[dev] < RING) {
        if (hipEventCreateWithFlags(&ring[dev][n[dev]], hipEventDisableTiming) != hipSuccess) return nullptr;
        return ring[dev][n[dev]++];
    }
    at[dev] = (at[dev] + 1) % RING;
    return ring[dev][at[dev]];
}

// dW = dY^T X: A(m = o, k = r) = dY[r][o], B(n = i, k = r) = X[r][i] — both k-major, read in place
struct Streams { void* ws; size_t ws_bytes; void* side_stream; void* side_ws; size_t side_ws_bytes; };
int grad_weight(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* out, int O, int I, int R, const Streams& g, hipStream_t main) {
    if (!g.side_stream) return gemm_layouts(dy, ldy, 1, x, ldx, 1, nullptr, 0, out, O, I, R, g.ws, g.ws_bytes, main);
    hipStream_t side = reinterpret_cast<hipStream_t>(g.side_stream);
    hipEvent_t ev = next_event();
    if (!ev) return (int)hipErrorOutOfMemory;
    if (hipError_t e = hipEventRecord(ev, main)) return (int)e;
    if (hipError_t e = hipStreamWaitEvent(side, ev, 0)) return (int)e;
    return gemm_layouts(dy, ldy, 1, x, ldx, 1, nullptr, 0, out, O, I, R, g.side_ws, g.side_ws_bytes, side);
}
int grad_weight(const float* dy, int64_t ldy, const float* x, int64_t ldx, float* out, int O, int I, int R, const hirest_train_block* b,
                const hirest_train_block_grads* g, hipStream_t main) {
    return grad_weight(dy, ldy, x, ldx, out, O, I, R, Streams{b->ws, b->ws_bytes, g->side_stream, g->side_ws, g->side_ws_bytes}, main);
}

// ---- precision 1: forward and dX products on split operands (three bf16 MFMAs per product, gemm_t128x3) ----
int gemm_x3(const hirest_bf16* A2, const hirest_bf16* W2, const float* bias, float* out, int M, int N, int K, int epi, void* part, hipStream_t s) {
    hirest_gemm_args a;
    a.struct_size = sizeof(a);
    a.A = A2; a.lda = 2 * (int64_t)K; a.W = W2; a.ldw = 2 * (int64_t)K; a.bias = bias; a.out = out; a.ldo = N;
    a.M = M; a.N = N; a.K = 2 * K; a.epilogue = epi; a.pos = nullptr; a.patches_per_frame = 0; a.aux0 = part; a.aux1 = nullptr;
    a.flags = HIREST_GEMM_X3 | HIREST_GEMM_X3_T128;
    return hirest_gemm_bf16(&a, s);
}
inline size_t alb(size_t bf16s) { return (bf16s + 127) & ~(size_t)127; }         // 256-B aligned bf16 slices

struct FwdX3 { size_t o, part, x2, cx2, aa2, hh2, wqkv2, wo2, w12, w22, total; };
FwdX3 plan_fwd_x3(const Shape& d) {
    FwdX3 r; size_t off = 0;
    auto f32 = [&off](size_t n) { size_t at = off; off += al(n) * 4; return at; };
    auto b16 = [&off](size_t n) { size_t at = off; off += alb(n) * 2; return at; };
    r.o = f32(d.R * d.W); r.part = f32(4 * d.R * d.W);
    r.x2 = b16(d.R * 2 * d.W); r.cx2 = b16(d.R * 2 * d.W); r.aa2 = b16(d.R * 2 * d.W); r.hh2 = b16(d.R * 2 * d.mlp);
    r.wqkv2 = b16(d.M3 * 2 * d.W); r.wo2 = b16(d.W * 2 * d.W); r.w12 = b16(d.mlp * 2 * d.W); r.w22 = b16(d.W * 2 * d.mlp);
    r.total = off;
    return r;
}
struct BwdX3 { size_t part, dy2, dhp2, do2, dqkv2, w2t, w1t, wot, wqkvt, total; };
BwdX3 plan_bwd_x3(const Shape& d, size_t base) {
    BwdX3 r; size_t off = base;
    auto f32 = [&off](size_t n) { size_t at = off; off += al(n) * 4; return at; };
    auto b16 = [&off](size_t n) { size_t at = off; off += alb(n) * 2; return at; };
    r.part = f32(4 * d.R * d.W);
    r.dy2 = b16(d.R * 2 * d.W); r.dhp2 = b16(d.R * 2 * d.mlp); r.do2 = b16(d.R * 2 * d.W); r.dqkv2 = b16(d.R * 2 * d.M3);
    r.w2t = b16(d.mlp * 2 * d.W); r.w1t = b16(d.W * 2 * d.mlp); r.wot = b16(d.W * 2 * d.W); r.wqkvt = b16(d.W * 2 * d.M3);
    r.total = off;
    return r;
}

struct Items { hirest_colsum_item* items; int32_t* n; int32_t max; };
int colsum_item(const Items& g, const float* x, int64_t ldx, int R, int C, float* out, const float* weight = nullptr, const int32_t* select = nullptr,
                int value = 0) {
    if (!g.items || !g.n || *g.n >= g.max) return HIREST_E_BADARG;
    hirest_colsum_item& it = g.items[(*g.n)++];
    it.x = x; it.row_weight = weight; it.row_select = select; it.out = out; it.ldx = ldx; it.R = R; it.C = C; it.select_value = value; it.reserved = 0;
    return 0;
}
int colsum_item(const hirest_train_block_grads* g, const float* x, int64_t ldx, int R, int C, float* out) {
    return colsum_item(Items{g->items, g->n_items, g->max_items}, x, ldx, R, C, out);
}

}  // namespace

extern "C" size_t hirest_train_block_forward_scratch_bytes(const hirest_train_block* b) {
    if (!block_ok(b)) return 0;
    const Shape d(b);
    if (b->precision == 1) return plan_fwd_x3(d).total;
    return al(d.R * d.W) * sizeof(float);                                        // o / y: the dense outputs in front of their dropout + add
}

extern "C" int hirest_train_block_forward(const hirest_train_block* b, void* scratch, size_t scratch_bytes, void* stream) {
    if (!block_ok(b) || !scratch || scratch_bytes < hirest_train_block_forward_scratch_bytes(b)) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Shape d(b);
    const int R = (int)d.R, W = (int)d.W, M3 = (int)d.M3, mlp = (int)d.mlp, dh = W / b->heads;
    const float scale = (float)pow((double)dh, -0.5);
    if (b->precision == 1) {
        if (W % 32 != 0 || mlp % 32 != 0 || W > 2048) return HIREST_E_SHAPE;
        const FwdX3 r = plan_fwd_x3(d);
        char* base = reinterpret_cast<char*>(scratch);
        auto H = [base](size_t at) { return reinterpret_cast<hirest_bf16*>(base + at); };
        float* o = reinterpret_cast<float*>(base + r.o);
        void* part = base + r.part;
        // the four weights in the operand format (they change every optimizer step): the caller's, or split here
        const hirest_bf16 *wqkv2 = b->wqkv2, *wo2 = b->wo2, *w12 = b->w12, *w22 = b->w22;
        if (!wqkv2 || !wo2 || !w12 || !w22) {
            CHECK(hirest_split2_bf16(b->wqkv, W, H(r.wqkv2), 2 * W, M3, W, 0, s));
            CHECK(hirest_split2_bf16(b->wo, W, H(r.wo2), 2 * W, W, W, 0, s));
            CHECK(hirest_split2_bf16(b->w1, W, H(r.w12), 2 * W, mlp, W, 0, s));
            CHECK(hirest_split2_bf16(b->w2, mlp, H(r.w22), 2 * mlp, W, mlp, 0, s));
            wqkv2 = H(r.wqkv2); wo2 = H(r.wo2); w12 = H(r.w12); w22 = H(r.w22);
        }
        const hirest_bf16* x2 = b->x2;
        if (!x2) { CHECK(hirest_split2_bf16(b->x, W, H(r.x2), 2 * W, R, W, 0, s)); x2 = H(r.x2); }
        CHECK(gemm_x3(x2, wqkv2, b->bqkv, b->qkv, R, M3, W, HIREST_EPI_BIAS_F32, nullptr, s));
        CHECK(hirest_attention_train_fwd_f32(b->qkv, b->P, b->cx, b->B, b->T, b->heads, dh, scale, -10000.0f, b->drop, b->seed_attn, s));
        CHECK(hirest_split2_bf16(b->cx, W, H(r.cx2), 2 * W, R, W, 0, s));
        CHECK(gemm_x3(H(r.cx2), wo2, b->bo, o, R, W, W, HIREST_EPI_BIAS_F32, nullptr, s));
        CHECK(hirest_dropout_add_f32(o, b->x, b->a_pre, d.R * d.W, b->drop, b->seed_ao, s));
        CHECK(hirest_layernorm_f32_split2(b->a_pre, W, nullptr, 0, b->ln1_g, b->ln1_b, b->ln_eps, b->aa, W, H(r.aa2), 2 * W, R, W, s));
        CHECK(gemm_x3(H(r.aa2), w12, b->b1, b->hpre, R, mlp, W, HIREST_EPI_BIAS_F32, nullptr, s));
        CHECK(hirest_act_f32(b->hpre, b->hh, d.R * d.mlp, 1, s));
        CHECK(hirest_split2_bf16(b->hh, mlp, H(r.hh2), 2 * mlp, R, mlp, 0, s));
        // the 3072-deep product: K slices on otherwise idle CUs (accumulate form onto zeros)
        if (hipError_t e = hipMemsetAsync(o, 0, (size_t)d.R * d.W * sizeof(float), s)) return (int)e;
        CHECK(gemm_x3(H(r.hh2), w22, b->b2, o, R, W, mlp, HIREST_EPI_BIAS_RESID_F32, part, s));
        CHECK(hirest_dropout_add_f32(o, b->aa, b->x_pre, d.R * d.W, b->drop, b->seed_out, s));
        CHECK(hirest_layernorm_f32_split2(b->x_pre, W, nullptr, 0, b->ln2_g, b->ln2_b, b->ln_eps, b->out, W, b->out2, 2 * W, R, W, s));
        return 0;
    }
    float* o = reinterpret_cast<float*>(scratch);
    CHECK(gemm(b->x, W, b->wqkv, W, b->bqkv, b->qkv, R, M3, W, b->ws, b->ws_bytes, s));
    CHECK(hirest_attention_train_fwd_f32(b->qkv, b->P, b->cx, b->B, b->T, b->heads, dh, scale, -10000.0f, b->drop, b->seed_attn, s));
    CHECK(gemm(b->cx, W, b->wo, W, b->bo, o, R, W, W, b->ws, b->ws_bytes, s));
    CHECK(hirest_dropout_add_f32(o, b->x, b->a_pre, d.R * d.W, b->drop, b->seed_ao, s));
    CHECK(hirest_layernorm(b->a_pre, W, nullptr, b->ln1_g, b->ln1_b, b->ln_eps, b->aa, W, 1, R, W, s));
    CHECK(gemm(b->aa, W, b->w1, W, b->b1, b->hpre, R, mlp, W, b->ws, b->ws_bytes, s));
    CHECK(hirest_act_f32(b->hpre, b->hh, d.R * d.mlp, 1, s));
    CHECK(gemm(b->hh, mlp, b->w2, mlp, b->b2, o, R, W, mlp, b->ws, b->ws_bytes, s));
    CHECK(hirest_dropout_add_f32(o, b->aa, b->x_pre, d.R * d.W, b->drop, b->seed_out, s));
    CHECK(hirest_layernorm(b->x_pre, W, nullptr, b->ln2_g, b->ln2_b, b->ln_eps, b->out, W, 1, R, W, s));
    return 0;
}

extern "C" size_t hirest_train_block_backward_scratch_bytes(const hirest_train_block* b) {
    if (!block_ok(b)) return 0;
    const Shape d(b);
    // dxp, dyx2, dy, da, dap, dyx1, do, dcx: [R, W] each;  dh, dhp: [R, mlp];  dS: [B, H, T, T];  dqkv: [R, 3 W]
    const size_t f32_part = (8 * al(d.R * d.W) + 2 * al(d.R * d.mlp) + al(d.PT) + al(d.R * d.M3)) * sizeof(float);
    return b->precision == 1 ? plan_bwd_x3(d, f32_part).total : f32_part;
}

extern "C" int hirest_train_block_backward(const hirest_train_block* b, const hirest_train_block_grads* g, void* stream) {
    if (!block_ok(b) || !g || g->struct_size != sizeof(*g) || !g->dout || !g->dx || !g->scratch ||
        g->scratch_bytes < hirest_train_block_backward_scratch_bytes(b))
        return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Shape d(b);
    const int R = (int)d.R, W = (int)d.W, M3 = (int)d.M3, mlp = (int)d.mlp, dh = W / b->heads;
    const float scale = (float)pow((double)dh, -0.5);
    const bool x3 = b->precision == 1;
    if (x3 && (W % 32 != 0 || mlp % 32 != 0)) return HIREST_E_SHAPE;
    char* sbase = reinterpret_cast<char*>(g->scratch);
    const BwdX3 q = plan_bwd_x3(d, (8 * al(d.R * d.W) + 2 * al(d.R * d.mlp) + al(d.PT) + al(d.R * d.M3)) * sizeof(float));
    auto H = [sbase](size_t at) { return reinterpret_cast<hirest_bf16*>(sbase + at); };
    void* part = sbase + q.part;
    // dX = dY W on split operands: A = split(dY), B = split(W^T) (hirest_split2_transposed_bf16); `resid` is copied into dx first and the
    // product accumulates onto it (resid itself may still be read by the side stream's dW product)
    auto dx_x3 = [&](const float* dY, int O, hirest_bf16* dY2, const float* Wt, int I, const hirest_bf16* given, hirest_bf16* mine, const float* resid,
                     float* dx) -> int {
        CHECK(hirest_split2_bf16(dY, O, dY2, 2 * O, R, O, 0, s));
        const hirest_bf16* WT2 = given;
        if (!WT2) { CHECK(hirest_split2_transposed_bf16(Wt, I, mine, 2 * O, O, I, s)); WT2 = mine; }
        if (!resid) return gemm_x3(dY2, WT2, nullptr, dx, R, I, O, HIREST_EPI_BIAS_F32, nullptr, s);
        if (hipError_t e = hipMemcpyAsync(dx, resid, (size_t)R * I * sizeof(float), hipMemcpyDeviceToDevice, s)) return (int)e;
        return gemm_x3(dY2, WT2, nullptr, dx, R, I, O, HIREST_EPI_BIAS_RESID_F32, part, s);
    };
    float* p = reinterpret_cast<float*>(g->scratch);
    auto take = [&p](int64_t n) { float* q = p; p += al((size_t)n); return q; };
    float *dxp = take(d.R * d.W), *dyx2 = take(d.R * d.W), *dyb = take(d.R * d.W), *da = take(d.R * d.W), *dap = take(d.R * d.W),
          *dyx1 = take(d.R * d.W), *dob = take(d.R * d.W), *dcx = take(d.R * d.W), *dhid = take(d.R * d.mlp), *dhp = take(d.R * d.mlp),
          *dS = take(d.PT), *dqkv = take(d.R * d.M3);
    const bool drop = b->drop != 0.0f;

    // output.LayerNorm, dropout(output.dense(...)) + aa
    CHECK(hirest_layernorm_bwd_f32(b->x_pre, g->dout, b->ln2_g, b->ln_eps, dxp, dyx2, R, W, s));
    CHECK(colsum_item(g, dyx2, W, R, W, g->g_ln2_g));
    CHECK(colsum_item(g, g->dout, W, R, W, g->g_ln2_b));
    const float* dy = dxp;                                                       // through dropout(y); the residual branch gets dxp as is
    if (drop) { CHECK(hirest_dropout_add_f32(dxp, nullptr, dyb, d.R * d.W, b->drop, b->seed_out, s)); dy = dyb; }
    CHECK(grad_weight(dy, W, b->hh, mlp, g->g_w2, W, mlp, R, b, g, s));
    CHECK(colsum_item(g, dy, W, R, W, g->g_b2));
    if (x3) CHECK(dx_x3(dy, W, H(q.dy2), b->w2, mlp, b->w2T2, H(q.w2t), nullptr, dhid));
    else CHECK(gemm_layouts(dy, W, 0, b->w2, mlp, 1, nullptr, 0, dhid, R, mlp, W, b->ws, b->ws_bytes, s));     // dX = dY W: B(n = i, k = o) = W[o][i]
    CHECK(hirest_act_bwd_f32(b->hpre, dhid, dhp, d.R * d.mlp, 1, s));
    CHECK(grad_weight(dhp, mlp, b->aa, W, g->g_w1, mlp, W, R, b, g, s));
    CHECK(colsum_item(g, dhp, mlp, R, mlp, g->g_b1));
    if (x3) CHECK(dx_x3(dhp, mlp, H(q.dhp2), b->w1, W, b->w1T2, H(q.w1t), dxp, da));
    else CHECK(gemm_layouts(dhp, mlp, 0, b->w1, W, 1, dxp, W, da, R, W, mlp, b->ws, b->ws_bytes, s));           // + the residual path
    // attention.output.LayerNorm, dropout(attention.output.dense(cx)) + x
    CHECK(hirest_layernorm_bwd_f32(b->a_pre, da, b->ln1_g, b->ln_eps, dap, dyx1, R, W, s));
    CHECK(colsum_item(g, dyx1, W, R, W, g->g_ln1_g));
    CHECK(colsum_item(g, da, W, R, W, g->g_ln1_b));
    const float* dO = dap;
    if (drop) { CHECK(hirest_dropout_add_f32(dap, nullptr, dob, d.R * d.W, b->drop, b->seed_ao, s)); dO = dob; }
    CHECK(grad_weight(dO, W, b->cx, W, g->g_wo, W, W, R, b, g, s));
    CHECK(colsum_item(g, dO, W, R, W, g->g_bo));
    if (x3) CHECK(dx_x3(dO, W, H(q.do2), b->wo, W, b->woT2, H(q.wot), nullptr, dcx));
    else CHECK(gemm_layouts(dO, W, 0, b->wo, W, 1, nullptr, 0, dcx, R, W, W, b->ws, b->ws_bytes, s));
    // self-attention
    CHECK(hirest_attention_train_bwd_f32(b->qkv, b->P, dcx, dS, dqkv, b->B, b->T, b->heads, dh, scale, b->drop, b->seed_attn, s));
    CHECK(grad_weight(dqkv, M3, b->x, W, g->g_wqkv, M3, W, R, b, g, s));
    CHECK(colsum_item(g, dqkv, M3, R, M3, g->g_bqkv));
    if (x3) CHECK(dx_x3(dqkv, M3, H(q.dqkv2), b->wqkv, W, b->wqkvT2, H(q.wqkvt), dap, g->dx));
    else CHECK(gemm_layouts(dqkv, M3, 0, b->wqkv, W, 1, dap, W, g->dx, R, W, M3, b->ws, b->ws_bytes, s));       // + the residual path
    return 0;
}

// ---- the backward below the encoder blocks: embeddings + fusion (train.py:_encoder_backward after its block loop, call for call) ----
namespace {
struct FusionPlan { size_t dxe, dx0, dyx, df, dv, dtn, tmp, dpre, time, da0, dxa, dyxa, dv0, dyxv, dt, total; };
FusionPlan plan_fusion(const hirest_train_fusion_bwd* f) {
    const size_t R = (size_t)f->B * f->T, W = f->W, E = f->E, A = f->asr_dim;
    FusionPlan r; size_t off = 0;
    auto take = [&off](size_t n) { size_t at = off; off += al(n) * sizeof(float); return at; };
    r.dxe = take(R * W); r.dx0 = take(R * W); r.dyx = take(R * W); r.df = take(R * E); r.dv = take(R * E); r.dtn = take((size_t)f->B * E);
    r.tmp = take(R * E); r.dpre = take(R * E); r.time = take(R); r.da0 = take(R * A); r.dxa = take(R * A); r.dyxa = take(R * A);
    r.dv0 = take(R * E); r.dyxv = take(R * E); r.dt = take((size_t)f->B * E);
    r.total = off;
    return r;
}
inline bool fusion_ok(const hirest_train_fusion_bwd* f) {
    return f && f->struct_size == sizeof(*f) && f->B > 0 && f->T > 0 && f->E > 0 && f->W > 0 && f->vis_dim > 0 && f->text_dim > 0 && f->asr_dim >= 0 &&
           f->T <= f->max_pos && f->E % 16 == 0 && f->W % 16 == 0 && f->vis_dim % 4 == 0 && f->text_dim % 4 == 0 && f->asr_dim % 4 == 0 &&
           (int64_t)f->T * f->W <= 0x7fffffff;
}
}  // namespace

extern "C" size_t hirest_train_fusion_backward_scratch_bytes(const hirest_train_fusion_bwd* f) { return fusion_ok(f) ? plan_fusion(f).total : 0; }

extern "C" int hirest_train_fusion_backward(const hirest_train_fusion_bwd* f, void* stream) {
    if (!fusion_ok(f) || !f->dx || !f->scratch || f->scratch_bytes < plan_fusion(f).total || !f->g_pos) return HIREST_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int B = f->B, T = f->T, E = f->E, W = f->W, A = f->asr_dim, R = B * T;
    const FusionPlan q = plan_fusion(f);
    char* base = reinterpret_cast<char*>(f->scratch);
    auto F = [base](size_t at) { return reinterpret_cast<float*>(base + at); };
    const Items it{f->items, f->n_items, f->max_items};
    const Streams st{f->ws, f->ws_bytes, f->side_stream, f->side_ws, f->side_ws_bytes};
    // embeddings: dropout, LayerNorm, position rows, word_embeddings (a Linear)
    const float* dxe = f->dx;
    if (f->drop != 0.0f) { CHECK(hirest_dropout_add_f32(f->dx, nullptr, F(q.dxe), (int64_t)R * W, f->drop, f->seed_emb, s)); dxe = F(q.dxe); }
    float* dx0 = F(q.dx0);
    CHECK(hirest_layernorm_bwd_f32(f->x0, dxe, f->emb_ln_g, 1e-12f, dx0, F(q.dyx), R, W, s));
    CHECK(colsum_item(it, F(q.dyx), W, R, W, f->g_emb_ln_g));
    CHECK(colsum_item(it, dxe, W, R, W, f->g_emb_ln_b));
    if (hipError_t e = hipMemsetAsync(f->g_pos, 0, (size_t)f->max_pos * W * sizeof(float), s)) return (int)e;
    CHECK(colsum_item(it, dx0, (int64_t)T * W, B, T * W, f->g_pos));              // rows t of every video add up: dx0 viewed as [B, T W]
    CHECK(grad_weight(dx0, W, f->f, E, f->g_w_emb, W, E, R, st, s));
    CHECK(colsum_item(it, dx0, W, R, W, f->g_b_emb));
    float* df = F(q.df);
    CHECK(gemm_layouts(dx0, W, 0, f->w_emb, E, 1, nullptr, 0, df, R, E, W, f->ws, f->ws_bytes, s));
    // fusion: f = (v * tn + asr + temporal) + mask_embed[moment_mask] (+ boundary_embed[boundary_mask])
    for (int k = 0; k < 2; ++k) CHECK(colsum_item(it, df, E, R, E, f->g_mask + (size_t)k * E, nullptr, f->mm32, k));
    if (f->boundary)
        for (int k = 0; k < 2; ++k) CHECK(colsum_item(it, df, E, R, E, f->g_bound + (size_t)k * E, nullptr, f->bm32, k));
    CHECK(hirest_joint_base_bwd_f32(df, f->v, f->tn, F(q.dv), F(q.dtn), B, T, E, s));
    // temporal embedding: Linear(1, E) -> tanh -> Linear(E, E) over the normalised time grid
    CHECK(grad_weight(df, E, f->tin, E, f->g_t2_w, E, E, R, st, s));
    CHECK(colsum_item(it, df, E, R, E, f->g_t2_b));
    CHECK(gemm_layouts(df, E, 0, f->t2_w, E, 1, nullptr, 0, F(q.tmp), R, E, E, f->ws, f->ws_bytes, s));
    CHECK(hirest_act_bwd_f32(f->tin, F(q.tmp), F(q.dpre), (int64_t)R * E, 3, s));
    CHECK(hirest_joint_time_grid_f32(f->n_valid, B, T, F(q.time), s));
    CHECK(colsum_item(it, F(q.dpre), E, R, E, f->g_t0_w, F(q.time)));
    CHECK(colsum_item(it, F(q.dpre), E, R, E, f->g_t0_b));
    if (A > 0) {                                                                  // asr_enc_layer: LayerNorm(asr_dim) -> Linear(asr_dim, E)
        CHECK(grad_weight(df, E, f->a0, A, f->g_asr1_w, E, A, R, st, s));
        CHECK(colsum_item(it, df, E, R, E, f->g_asr1_b));
        CHECK(gemm_layouts(df, E, 0, f->asr1_w, A, 1, nullptr, 0, F(q.da0), R, A, E, f->ws, f->ws_bytes, s));
        CHECK(hirest_layernorm_bwd_f32(f->asr2, F(q.da0), f->asr0_g, 1e-5f, F(q.dxa), F(q.dyxa), R, A, s));
        CHECK(colsum_item(it, F(q.dyxa), A, R, A, f->g_asr0_g));
        CHECK(colsum_item(it, F(q.da0), A, R, A, f->g_asr0_b));
    }
    // normalize_video (LayerNorm) <- clip_g_map;  tn = t / |t| <- clip_g_map_text
    CHECK(hirest_layernorm_bwd_f32(f->v0, F(q.dv), f->norm_g, 1e-12f, F(q.dv0), F(q.dyxv), R, E, s));
    CHECK(colsum_item(it, F(q.dyxv), E, R, E, f->g_norm_g));
    CHECK(colsum_item(it, F(q.dv), E, R, E, f->g_norm_b));
    CHECK(grad_weight(F(q.dv0), E, f->vis2, f->vis_dim, f->g_vis_w, E, f->vis_dim, R, st, s));
    CHECK(colsum_item(it, F(q.dv0), E, R, E, f->g_vis_b));
    CHECK(hirest_l2norm_bwd_f32(f->t, F(q.dtn), F(q.dt), B, E, s));
    CHECK(grad_weight(F(q.dt), E, f->text, f->text_dim, f->g_text_w, E, f->text_dim, B, st, s));
    CHECK(colsum_item(it, F(q.dt), E, B, E, f->g_text_b));
    return 0;
}

