// Joint-model (MomentModel) kernels: the moment-retrieval / moment-segmentation heads of
// /root/reference/modeling.py:155-474 over precomputed per-second frame features.
//
// Everything here is fp32 end to end.  The reference runs this model in fp32 and its outputs are frame
// INDICES (argmax of start/end logits, the iterative threshold walk); bf16 operands would move logits by
// ~1e-2 and flip near-ties, so this path uses the exact-fp32 matrix instructions
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, 1/16 of the bf16 rate) — at 10-90 GFLOP per video the
// whole model is still a few hundred microseconds.
#include "common.h"
#include <atomic>
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ resid[m][n]) (+ periodic[(m % period)][n])
// 64x64 tile, 4 waves (2x2) of one 32x32 MFMA tile each, K staged 16 deep through padded LDS.
// Operands are swapped (a = W, b = A) so a lane owns 4 consecutive columns of one row.
// ---------------------------------------------------------------------------------------------
struct GemmF {
    const float* A; int64_t lda;
    const float* W; int64_t ldw;
    const float* bias;
    const float* resid; int64_t ldr;
    const float* periodic; int period;
    float* out; int64_t ldo;
    int M, N, K, act;   // act: 0 none, 1 gelu (erf), 2 tanh, 3 quick-gelu
    float* ws;          // gemm_f32_kernel only: not NULL = blockIdx.z is a K quarter whose partial sum goes to ws[z][M][N] (split form)
};

constexpr int FK = 32, FLD = FK + 1;      // K is staged 32 deep; K itself only has to be a multiple of 16 (zero fill)

// The one epilogue of every fp32 GEMM kernel below (same expressions -> same bits): four consecutive columns n .. n + 3 of row m.
// Its operands are fetched apart from their use — unconditionally, on clamped indices — so that a kernel can have all of them
// travelling at once, or from its first instruction: fetched where they are used, every `if (p.bias) v += load` is a memory round
// trip of its own at the very end of a kernel (two per tile: 1-2 us of a 5-us decode GEMM).
struct Epi4 { f32x4 bias, resid, periodic; };
__device__ __forceinline__ Epi4 epilogue_fetch4(const GemmF& p, int m, int n) {      // m, n may lie outside: the values are then unused
    Epi4 e;
    const int mc = m < p.M ? m : p.M - 1, nc = n + 4 <= p.N ? n : p.N - 4;            // N % 4 == 0
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    e.bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : zero;
    e.resid = p.resid ? *reinterpret_cast<const f32x4*>(p.resid + (int64_t)mc * p.ldr + nc) : zero;
    e.periodic = p.periodic ? *reinterpret_cast<const f32x4*>(p.periodic + (int64_t)(mc % p.period) * p.N + nc) : zero;
    return e;
}
__device__ __forceinline__ f32x4 epilogue_apply4(const GemmF& p, f32x4 v, const Epi4& e, int m, int n, bool store = true) {
    if (p.bias) v += e.bias;
    if (p.act == 1) {
#pragma unroll
        for (int e_ = 0; e_ < 4; ++e_) v[e_] = 0.5f * v[e_] * (1.0f + erff(v[e_] * 0.70710678118654752440f));
    } else if (p.act == 2) {
#pragma unroll
        for (int e_ = 0; e_ < 4; ++e_) v[e_] = tanhf(v[e_]);
    } else if (p.act == 3) {                      // QuickGELU (model.py:175-177), the fp32 reference-precision towers
#pragma unroll
        for (int e_ = 0; e_ < 4; ++e_) v[e_] = v[e_] * (1.0f / (1.0f + expf(-1.702f * v[e_])));
    }
    if (p.resid) v += e.resid;
    if (p.periodic) v += e.periodic;
    if (store) *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.ldo + n) = v;
    return v;
}

// AK / WK: the operand is stored k-major — A(m, k) at A[k * lda + m], W(n, k) at W[k * ldw + n] — as the backward pass of a linear
// layer has them (dX = dY W: W(n, k) = W[k][n];  dW = dY^T X: both).  Such an operand is read along its contiguous dimension
// (4 consecutive rows of one k per thread) and transposed on its way into the same LDS image, so the products, their order and
// the bits are those of the GEMM on transposed copies, without the copies (the training step made 37 of them per iteration).
template <bool AK, bool WK>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(GemmF p) {
    // The four blocks that share a CU start together and would run their memory and MFMA phases in step (the matrix pipe idles while
    // all of them store / synchronise / read): groups of 32 consecutive blocks start 0 / 512 / 1024 / 1536 cycles late.  Grids that
    // divide evenly over the CUs gain 10-17 % (1024 x 4096 x 768: 75 -> 62 us, 4096 x 3072 x 768: 189 -> 171), M = 1500 nothing.
    {
        const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        for (int i = (bid >> 5) & 3; i > 0; --i) __builtin_amdgcn_s_sleep(8);
    }
    constexpr int TLD = 36;                                  // tile row stride in floats: rows 16-B aligned, 8 consecutive rows on distinct banks
    __shared__ __attribute__((aligned(16))) float As[2][64 * TLD];   // two slabs: the next one is stored while this one is multiplied
    __shared__ __attribute__((aligned(16))) float Ws[2][64 * TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M0 = blockIdx.y * 64, N0 = blockIdx.x * 64;
    // staging: thread -> (row = tid/4, 8 consecutive k = (tid%4)*8); the next 32-deep slab is loaded into registers while
    // the current one is multiplied — these GEMMs run as one or two waves of blocks (M = 25 ... 1500), so the global
    // latency of a synchronous load per slab (44 us for K = 768) was the whole kernel time
    const int srow = tid >> 2, sk = (tid & 3) * 8;
    int gm = M0 + srow; gm = gm < p.M ? gm : p.M - 1;
    int gn = N0 + srow; gn = gn < p.N ? gn : p.N - 1;
    // k-major operand: thread -> (k = tid / 16 (+ 16 h), rows 4 (tid % 16) .. + 3 of the tile); rows past the end re-read the last four
    const int tk = tid >> 4, tc = (tid & 15) * 4;
    int cm = M0 + tc; cm = cm + 4 <= p.M ? cm : p.M - 4;
    int cn = N0 + tc; cn = cn + 4 <= p.N ? cn : p.N - 4;
    const float* ap = AK ? p.A + cm : p.A + (int64_t)gm * p.lda + sk;
    const float* wp = WK ? p.W + cn : p.W + (int64_t)gn * p.ldw + sk;
    // Summation order (shared with gemm_f32_skinny_kernel below, so that a row's result does not depend on how many rows the call
    // has): K is cut into four contiguous quarters of ceil(nslab / 4) 32-deep slabs, quarter u gives partial sum p_u (slabs in
    // ascending order; inside a slab MFMA step j pairs k0 + j (lanes < 32) with k0 + 16 + j (lanes >= 32)), and the result is
    // ((p0 + p1) + p2) + p3.  Here the quarters are walked one after the other, so memory is read front to back and only one
    // accumulator block is live (four live ones cost 13 % on moment retrieval / segmentation: occupancy).
    // LDS slab image: row r holds its 32 k's in natural order as eight 16-B chunks, chunk c at position c ^ ((r >> 4) & 3).  The step
    // that pairs k0 + j with k0 + 16 + j reads float j of chunks 0..3 (lanes < 32) / 4..7 (lanes >= 32): four ds_read_b128 per operand
    // and slab instead of sixteen ds_read_b32 (the scalar image at a 33-float stride spent 25 % of its LDS cycles on bank conflicts),
    // a row-major operand is stored with two ds_write_b128 per thread, and the XOR keeps the k-major operand's scalar stores (16 lanes
    // = 16 row quadruples at one k) on distinct banks.
    const int frow_a = wm * 32 + (lane & 31), frow_w = wn * 32 + (lane & 31);
    const int fsw_a = (frow_a >> 4) & 3, fsw_w = (frow_w >> 4) & 3, fhalf = 4 * (lane >> 5);
    const int ssw = (srow >> 4) & 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // The loads are written as asm so that the WAITS are ours: the compiler's own counter tracking turns the two-slabs-ahead prefetch
    // into vmcnt(0) at the loop header (it cannot prove which set is older across the back edge).  A set is touched again only through
    // landed<N>(), whose "+v" operands tie it to the s_waitcnt; every load is unconditional, on a clamped k (what lies past K becomes
    // zero when the slab is stored to the LDS; K % 4 == 0 for a row-major operand: a 4-vector is inside or outside as a whole).
    auto gload = [](const float* ptr) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(ptr) : "memory"); return v; };
    auto fetch = [&](int k0, f32x4 (&av)[2], f32x4 (&wv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = k0 + sk + 4 * h < p.K ? k0 + 4 * h : p.K - 4 - sk;          // (ap / wp already point at column sk)
            const int kc = k0 + tk + 16 * h < p.K ? k0 + tk + 16 * h : p.K - 1;        // k-major: one k per thread
            av[h] = gload(AK ? ap + (int64_t)kc * p.lda : ap + kr);
            wv[h] = gload(WK ? wp + (int64_t)kc * p.ldw : wp + kr);
        }
    };
    const int nslab = (p.K + FK - 1) / FK;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 av[2], wv[2];
    const int quarter = (nslab + 3) >> 2;
    // split form (few tiles for the CUs: a 768-wide layer over 1500 rows is 288): one block per (tile, quarter), the partial sums
    // meet in gemm_f32_quarters_kernel in the same order ((p0 + p1) + p2) + p3 — same bits
    const int ub = p.ws ? (int)blockIdx.z : 0, ue = p.ws ? ub + 1 : 4;
    const int s_first = ub * quarter;
    const int s_last = p.ws ? ((ub + 1) * quarter < nslab ? (ub + 1) * quarter : nslab) : nslab;   // no slab of another block's quarter is fetched
    // Software pipeline of a wave, one s_barrier per slab:
    //   global -> registers: slab s + 2 is requested at the top of step s, right after slab s + 1 has left the registers for the LDS,
    //   and has the whole step to arrive; registers -> LDS one slab ahead (buffers 0 / 1 alternate); LDS -> fragment registers half a
    //   slab ahead: the chunks of slab s + 1 replace those of slab s as soon as its MFMAs have issued, so the ds_read latency lies
    //   under the second half of the MFMAs instead of in front of every group of four.  (Two register sets, i.e. two slabs in flight,
    //   do not fit 128 registers without spills, and a spill reload's vmcnt(0) drains the prefetch.)
    // Before (LDS single-buffered, two __syncthreads — whose fence also waits for the prefetch — per slab, fragments read in front of
    // every four MFMAs) the memory skeleton alone took 50 us and the MFMAs alone 68 us of an 85-us launch (M 1500, N 3072, K 768):
    // they ran one after the other.
    // Hazards: buffer b is stored for slab s + 1 at the top of step s; its previous content (slab s - 1) was read as fragments during
    // step s - 2 and those reads were waited for (lgkmcnt(0)) before the barrier of step s - 1, which every wave has passed.  The
    // barrier of step s publishes the stores (each wave's lgkmcnt(0) precedes it) to the fragment reads that follow it.
    auto lds_done = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto landed = [](f32x4 (&ar)[2], f32x4 (&wr)[2]) {       // (the "+v" operands tie the registers to the wait)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar[0]), "+v"(ar[1]), "+v"(wr[0]), "+v"(wr[1]) : : "memory");
    };
    auto store = [&](int slab_index, int buf, f32x4 (&ar)[2], f32x4 (&wr)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool in = slab_index * FK + sk + 4 * h < p.K, ink = slab_index * FK + tk + 16 * h < p.K;
            if (!(AK ? ink : in)) ar[h] = zero;
            if (!(WK ? ink : in)) wr[h] = zero;
            if (!AK) *reinterpret_cast<f32x4*>(&As[buf][srow * TLD + 4 * (((sk >> 2) + h) ^ ssw)]) = ar[h];
            if (!WK) *reinterpret_cast<f32x4*>(&Ws[buf][srow * TLD + 4 * (((sk >> 2) + h) ^ ssw)]) = wr[h];
#pragma unroll
            for (int e = 0; e < 4; ++e) {                     // k-major: k = tk + 16 h of rows tc .. tc + 3
                const int kk = tk + 16 * h, r = tc + e;
                if (AK) As[buf][r * TLD + 4 * ((kk >> 2) ^ ((r >> 4) & 3)) + (kk & 3)] = ar[h][e];
                if (WK) Ws[buf][r * TLD + 4 * ((kk >> 2) ^ ((r >> 4) & 3)) + (kk & 3)] = wr[h][e];
            }
        }
    };
    f32x4 aq[4], wq[4];
    auto frag = [&](int buf, int c) {
        aq[c] = *reinterpret_cast<const f32x4*>(&As[buf][frow_a * TLD + 4 * ((fhalf + c) ^ fsw_a)]);
        wq[c] = *reinterpret_cast<const f32x4*>(&Ws[buf][frow_w * TLD + 4 * ((fhalf + c) ^ fsw_w)]);
    };
    f32x16 part;
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e] = 0.f;
    int u = ub, sl = s_first;
    auto close_quarter = [&]() {                             // partial sum u is complete: acc = ((p0 + p1) + p2) + p3 as they come
        if (u == ub) acc = part;
        else {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += part[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) part[e] = 0.f;
        ++u;
    };
    // (the prefetch is issued unconditionally — past the block's last slab it re-reads that slab — so that the registers always belong
    //  to exactly one outstanding request)
    auto fetch_slab = [&](int s_) { fetch((s_ < s_last ? s_ : s_last - 1) * FK, av, wv); };
    if (s_first < s_last) {
        fetch_slab(s_first);
        landed(av, wv);
        store(s_first, 0, av, wv);
        fetch_slab(s_first + 1);
        lds_done();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) frag(0, c);
    }
    auto step = [&](int cur) {                               // buffer `cur` holds slab sl, the registers slab sl + 1 (in flight)
        const bool more = sl + 1 < s_last;
        landed(av, wv);
        if (more) store(sl + 1, cur ^ 1, av, wv);
        fetch_slab(sl + 2);
        lds_done();                                          // this slab's fragments (read during the previous step) and the stores above
#pragma unroll
        for (int j = 0; j < 8; ++j) part = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j >> 2][j & 3], aq[j >> 2][j & 3], part, 0, 0, 0);
        __builtin_amdgcn_s_barrier();
        if (more) { frag(cur ^ 1, 0); frag(cur ^ 1, 1); }
#pragma unroll
        for (int j = 8; j < 16; ++j) part = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j >> 2][j & 3], aq[j >> 2][j & 3], part, 0, 0, 0);
        if (more) { frag(cur ^ 1, 2); frag(cur ^ 1, 3); }
        ++sl;
        if (sl == (u + 1) * quarter || sl == nslab) close_quarter();
    };
#pragma unroll 1
    while (sl + 1 < s_last) {                                // slabs in pairs: the buffer numbers are literals
        step(0);
        step(1);
    }
    if (sl < s_last) step(0);
    if (s_first < s_last) landed(av, wv);                    // the surplus prefetch owns the registers until it lands
    while (u < ue) close_quarter();                          // quarters without a slab (K < 4 slabs) still take their place in the sum: + 0
    if (p.ws) {                                              // split form: the raw partial sum of quarter ub
        const int m = M0 + wm * 32 + (lane & 31);
        if (m >= p.M) return;
        float* wrow_out = p.ws + ((int64_t)ub * p.M + m) * p.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = N0 + wn * 32 + 8 * g + 4 * (lane >> 5);
            if (n < p.N) *reinterpret_cast<f32x4*>(wrow_out + n) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
        return;
    }
    const int m = M0 + wm * 32 + (lane & 31);
    Epi4 ep[4];                                              // the four column groups' operands in one burst
#pragma unroll
    for (int g = 0; g < 4; ++g) ep[g] = epilogue_fetch4(p, m, N0 + wn * 32 + 8 * g + 4 * (lane >> 5));
    if (m >= p.M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = N0 + wn * 32 + 8 * g + 4 * (lane >> 5);
        if (n >= p.N) continue;
        epilogue_apply4(p, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]}, ep[g], m, n);
    }
}

// The same 64x64 tile with its operands on an LDS-DMA ring (round 5), for problems of dozens of tiles per CU (the fp32 towers' layers).  A slab
// (64 A rows + 64 W rows x 128 B = 16 KB) is 16 LDS-DMA pieces of 8 whole lines, four per wave, DEPTH - 1 slabs ahead (3 of 4 slots, two blocks
// per CU), ONE bare barrier per slab:
//     wait own pieces of slab s (counted vmcnt) | lgkmcnt(0), s_barrier | refill slab s - 1's slot with slab s + DEPTH - 1 | 8 ds_read_b128 | 16 MFMA
// RAW: every wave waited for its own pieces of slab s before the barrier.  WAR: a wave's fragment reads of slab s - 1 retired (lgkmcnt(0))
// before it arrived at the barrier of step s, after which the slot is refilled.  Same products in the same order as gemm_f32_kernel (four K
// quarters walked one after the other, MFMA step j pairs k0 + j with k0 + 16 + j, ((p0 + p1) + p2) + p3) -> same bits.  Row-major operands,
// K % 32 == 0 (the DMA moves whole 128-B row pieces); everything else stays with the kernel above.
// LDS image: row r of an operand block at 128 r, global chunk c at position c ^ ((r >> 1) & 7): the 16-lane groups of ds_read_b128 reading one
// chunk of consecutive rows fall on 16 distinct 16-B slots.
template <int DEPTH>
__global__ __launch_bounds__(256, DEPTH <= 4 ? 2 : 1) void gemm_f32_ring_kernel(GemmF p) {
    constexpr int SLAB = 16384, WOFF = 8192, L = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int M0 = blockIdx.y * 64, N0 = blockIdx.x * 64;
    // this wave's four pieces of a slab: waves 0 / 1 the A rows 0-31 / 32-63, waves 2 / 3 the W rows
    const char* src[L];
    int dst[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int r = 32 * (wave & 1) + 8 * j + (lane >> 3);          // row inside the operand's 64-row block
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        if (wave < 2) { int gm = M0 + r; gm = gm < p.M ? gm : p.M - 1; src[j] = reinterpret_cast<const char*>(p.A + (int64_t)gm * p.lda) + 16 * chunk; }
        else { int gn = N0 + r; gn = gn < p.N ? gn : p.N - 1; src[j] = reinterpret_cast<const char*>(p.W + (int64_t)gn * p.ldw) + 16 * chunk; }
        dst[j] = (wave < 2 ? 0 : WOFF) + (32 * (wave & 1) + 8 * j) * 128;
    }
    const int nslab = p.K / FK;
    auto dma = [&](int s) {                                  // slab s -> ring slot s % DEPTH
        char* slot = smem + (s % DEPTH) * SLAB;
#pragma unroll
        for (int j = 0; j < L; ++j) glds16(src[j] + (int64_t)s * (FK * 4), slot + dst[j]);
    };
#pragma unroll 1
    for (int s = 0; s < DEPTH - 1 && s < nslab; ++s) dma(s);
    const int row = lane & 31, half = lane >> 5;
    const int ra = wm * 32 + row, rw = wn * 32 + row;
    int fa[4], fw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        fa[c] = ra * 128 + (((4 * half + c) ^ ((ra >> 1) & 7)) << 4);
        fw[c] = WOFF + rw * 128 + (((4 * half + c) ^ ((rw >> 1) & 7)) << 4);
    }
    f32x16 acc, part;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; part[e] = 0.f; }
    const int quarter = (nslab + 3) >> 2;
    int u = 0;
    auto close_quarter = [&]() {                             // partial sum u is complete: acc = ((p0 + p1) + p2) + p3 as they come
        if (u == 0) acc = part;
        else {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += part[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) part[e] = 0.f;
        ++u;
    };
#pragma unroll 1
    for (int s = 0; s < nslab; ++s) {
        // slabs s .. min(s + DEPTH - 2, nslab - 1) are in flight, in order: slab s has landed once at most DEPTH - 2 slabs are pending
        if (nslab - s >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + DEPTH - 1 < nslab) dma(s + DEPTH - 1);
        const char* slot = smem + (s % DEPTH) * SLAB;
        f32x4 aq[4], wq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            aq[c] = *reinterpret_cast<const f32x4*>(slot + fa[c]);
            wq[c] = *reinterpret_cast<const f32x4*>(slot + fw[c]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) part = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j >> 2][j & 3], aq[j >> 2][j & 3], part, 0, 0, 0);
        if (s + 1 == (u + 1) * quarter || s + 1 == nslab) close_quarter();
    }
    while (u < 4) close_quarter();                           // quarters without a slab (K < 4 slabs) still take their place in the sum: + 0
    const int m = M0 + wm * 32 + row;
    Epi4 ep[4];                                              // the four column groups' operands in one burst
#pragma unroll
    for (int g = 0; g < 4; ++g) ep[g] = epilogue_fetch4(p, m, N0 + wn * 32 + 8 * g + 4 * half);
    if (m >= p.M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = N0 + wn * 32 + 8 * g + 4 * half;
        if (n >= p.N) continue;
        epilogue_apply4(p, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]}, ep[g], m, n);
    }
}

template <int DEPTH>
int launch_ring(const GemmF& p, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_f32_ring_kernel<DEPTH>;
    if (int e = hirest_configure(kern, DEPTH * 16384, cfg)) return e;
    hipLaunchKernelGGL(kern, dim3((p.N + 63) / 64, (p.M + 63) / 64), dim3(256), DEPTH * 16384, s, p);
    return hirest_launch_status();
}

// second half of the split form: out = epilogue(((p0 + p1) + p2) + p3), four columns per thread
__global__ __launch_bounds__(256) void gemm_f32_quarters_kernel(GemmF p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N >> 2;
    if (i >= (int64_t)p.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
    const Epi4 ep = epilogue_fetch4(p, m, n);
    const int64_t q = (int64_t)p.M * p.N, o = (int64_t)m * p.N + n;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(p.ws + o), p1 = *reinterpret_cast<const f32x4*>(p.ws + q + o);
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(p.ws + 2 * q + o), p3 = *reinterpret_cast<const f32x4*>(p.ws + 3 * q + o);
    epilogue_apply4(p, ((p0 + p1) + p2) + p3, ep, m, n);
}

// Few rows (M <= 256: beam-search decoding, 5-25 rows per step): the 64x64 kernel above runs as a handful of blocks whose K loop
// is a chain of global-load latencies.  Here a block owns a 32x32 output tile and its four waves split K (wave w multiplies the
// w-th quarter of the 32-deep slabs), so 24-954 blocks x 4 waves work on a problem at once; operands go straight from global memory
// into registers (lane = row, 16 consecutive k per half-wave, three slabs in flight — no LDS, no barrier in the loop).  MFMA step
// j of a slab pairs k0 + j (lanes < 32) with k0 + 16 + j (lanes >= 32).  The four partial tiles are added in wave order through
// LDS, so the result does not depend on timing.
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(GemmF p) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar loop control
    const int row = lane & 31, khalf = lane >> 5;
    const int M0 = blockIdx.y * 32, N0 = blockIdx.x * 32;
    int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1;
    int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1;
    const float* ap = p.A + (int64_t)gm * p.lda + 16 * khalf;
    const float* wp = p.W + (int64_t)gn * p.ldw + 16 * khalf;
    const int nslab = (p.K + FK - 1) / FK;
    const int quarter = (nslab + 3) >> 2, first = wave * quarter;      // this wave's K quarter (see gemm_f32_kernel)
    int cnt = nslab - first; cnt = cnt < 0 ? 0 : (cnt > quarter ? quarter : cnt);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, f32x4 (&a)[4], f32x4 (&w)[4]) {
        const int k0 = (first + s) * FK;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool in = s < cnt && k0 + 16 * khalf + 4 * h < p.K;     // K % 4 == 0: whole 4-vectors
            a[h] = in ? *reinterpret_cast<const f32x4*>(ap + k0 + 4 * h) : zero;
            w[h] = in ? *reinterpret_cast<const f32x4*>(wp + k0 + 4 * h) : zero;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mm = [&](const f32x4 (&a)[4], const f32x4 (&w)[4]) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[h][e], a[h][e], acc, 0, 0, 0);
    };
    f32x4 a0[4], w0[4], a1[4], w1[4], a2[4], w2[4];
    load(0, a0, w0); load(1, a1, w1); load(2, a2, w2);
    for (int s = 0; s < cnt; s += 3) {
        mm(a0, w0); load(s + 3, a0, w0);
        if (s + 1 < cnt) { mm(a1, w1); load(s + 4, a1, w1); }
        if (s + 2 < cnt) { mm(a2, w2); load(s + 5, a2, w2); }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = ((acc[e] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane];
    const int m = M0 + row;
    Epi4 ep[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ep[g] = epilogue_fetch4(p, m, N0 + 8 * g + 4 * khalf);
    if (m >= p.M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = N0 + 8 * g + 4 * khalf;
        if (n >= p.N) continue;
        epilogue_apply4(p, f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]}, ep[g], m, n);
    }
}

// At most 32 rows (one beam-search step: 5 videos x 3-5 beams).  Two things bound the kernel above there.  (i) ONE dependent chain
// per wave: v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles, so a 192-deep K quarter is 6144 cycles, on 24 blocks.
// v_mfma_f32_16x16x4_f32 adds its four k in slot order as one fused chain (tools/probes/fma_order_probe.hip: bit-identical to the
// 32x32x2 stream when slots 0..3 carry k0+j, k0+16+j, k0+j+1, k0+16+j+1) at 11 cycles per k, on 16-column tiles: 48-1908 blocks.
// (ii) lane = row operand loads touch 16-32 cache lines per instruction for 16 B each, and the vector L1 serves about one line per
// four cycles whatever is used of it: the LM head (94 MB of weights) ran at 1.2-2.3 TB/s.  Here every operand slab (rows x 128 B)
// goes global -> LDS by LDS-DMA as whole lines (8 rows per wave-instruction) into a WAVE-PRIVATE ring of DEPTH slabs — wave w owns
// K quarter w of the block's column tiles, so nothing is shared and the loop has no barrier, only counted vmcnt waits — and the
// fragments are read back with ds_read_b128 (chunk position XOR-swizzled by row pair: conflict-free for the real b128 lane groups).
// MT row tiles x NT column tiles of 16 per wave; chains are independent, fragments shared.
template <int MT, int NT, int DEPTH>
__global__ __launch_bounds__(256) void gemm_f32_m16_kernel(GemmF p) {
    constexpr int TILES = MT + NT, SLAB = TILES * 2048, L = 2 * TILES;          // bytes per slab, LDS-DMA instructions per slab
    extern __shared__ __attribute__((aligned(16))) char smem[];                 // 4 waves x DEPTH x SLAB, then the reduction buffer
    f32x4 (*red)[MT * NT][64] = reinterpret_cast<f32x4 (*)[MT * NT][64]>(smem + 4 * DEPTH * SLAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: scalar loop control
    const int idx = lane & 15, slot = lane >> 4;
    const int N0 = blockIdx.x * 16 * NT, M0 = blockIdx.y * 16 * MT;
    const bool odd = slot >> 1;
    char* ring = smem + wave * DEPTH * SLAB;
    auto swz = [](int row) { const int pr = (row >> 1) & 7; return pr ^ ((((pr >> 1) ^ (pr >> 2)) & 1) << 2); };
    // LDS-DMA: instruction j of a slab moves rows 8 j .. 8 j + 7 of the stacked tiles (A row tiles first, then W column tiles);
    // lane = (row r = lane / 8, LDS chunk position q = lane % 8) fetches global chunk q ^ swz(row in its tile)
    const char* src[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int trow = 8 * (j & 1) + (lane >> 3), tile = j >> 1;
        const int chunk = (lane & 7) ^ swz(trow);
        if (tile < MT) {
            int gm = M0 + 16 * tile + trow; gm = gm < p.M ? gm : p.M - 1;
            src[j] = reinterpret_cast<const char*>(p.A + (int64_t)gm * p.lda) + 16 * chunk;
        } else {
            int gn = N0 + 16 * (tile - MT) + trow; gn = gn < p.N ? gn : p.N - 1;
            src[j] = reinterpret_cast<const char*>(p.W + (int64_t)gn * p.ldw) + 16 * chunk;
        }
    }
    Epi4 ep[NT][MT];                                                    // wave 0 stores: its epilogue operands start travelling now
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) ep[u][t] = epilogue_fetch4(p, M0 + 16 * t + idx, N0 + 16 * u + 4 * slot);
    }
    const int nslab = p.K / FK;                                         // K % 32 == 0 (launcher)
    const int quarter = (nslab + 3) >> 2, first = wave * quarter;      // this wave's K quarter (see gemm_f32_kernel)
    int cnt = nslab - first; cnt = cnt < 0 ? 0 : (cnt > quarter ? quarter : cnt);
    auto dma = [&](int s, int ringslot) {                               // slab s of this wave's quarter (clamped: always a valid address)
        const int64_t koff = (int64_t)(first + (s < cnt ? s : cnt - 1)) * (FK * 4);
        char* dst = ring + ringslot * SLAB;
#pragma unroll
        for (int j = 0; j < L; ++j) glds16(src[j] + koff, dst + j * 1024);
    };
    // fragment addresses: row idx of a tile, chunk 4 (slot & 1) + c at position chunk ^ swz(idx)
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = idx * 128 + (((4 * (slot & 1) + c) ^ swz(idx)) << 4);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](int ringslot) {
        const char* base = ring + ringslot * SLAB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 aq[MT], wq[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) aq[t] = *reinterpret_cast<const f32x4*>(base + t * 2048 + fo[c]);
#pragma unroll
            for (int u = 0; u < NT; ++u) wq[u] = *reinterpret_cast<const f32x4*>(base + (MT + u) * 2048 + fo[c]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {                               // instruction i = 2 c + e: slots carry k0 + 2 i + (slot >> 1) + 16 (slot & 1)
                float a[MT], w[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = odd ? aq[t][2 * e + 1] : aq[t][2 * e];
#pragma unroll
                for (int u = 0; u < NT; ++u) w[u] = odd ? wq[u][2 * e + 1] : wq[u][2 * e];
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u], a[t], acc[u][t], 0, 0, 0);
            }
        }
    };
    if (cnt > 0) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) dma(u, u);
        int rs = 0;                                                     // ring slot of the slab being multiplied
        for (int s = 0; s < cnt - DEPTH; ++s) {                         // steady state: DEPTH slabs in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * L) : "memory");
            mm(rs);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot's fragment reads have returned before it is refilled
            dma(s + DEPTH, rs);
            rs = rs + 1 == DEPTH ? 0 : rs + 1;
        }
        const int base = cnt > DEPTH ? cnt - DEPTH : 0;                 // drain: DEPTH slabs outstanding, the first cnt - base of them wanted
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (base + u < cnt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1 - u) * L) : "memory");
                mm(rs);
                rs = rs + 1 == DEPTH ? 0 : rs + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (clamped duplicates of a short quarter)
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) red[wave - 1][u * MT + t][lane] = acc[u][t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int n = N0 + 16 * u + 4 * slot;                           // D: column (row m of the output) = lane % 16, rows n = 4 (lane / 16) + r
        if (n >= p.N) continue;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = M0 + 16 * t + idx;
            if (m >= p.M) continue;
            epilogue_apply4(p, ((acc[u][t] + red[0][u * MT + t][lane]) + red[1][u * MT + t][lane]) + red[2][u * MT + t][lane], ep[u][t], m, n);
        }
    }
}

template <int MT, int NT, int DEPTH>
int launch_m16(const GemmF& p, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_f32_m16_kernel<MT, NT, DEPTH>;
    constexpr int LDS = 4 * DEPTH * (MT + NT) * 2048 + 3 * MT * NT * 1024;
    static_assert(LDS <= 160 * 1024, "ring does not fit the LDS");
    if (int e = hirest_configure(kern, LDS, cfg)) return e;
    hipLaunchKernelGGL(kern, dim3((p.N + 16 * NT - 1) / (16 * NT), (p.M + 16 * MT - 1) / (16 * MT)), dim3(256), LDS, s, p);
    return hirest_launch_status();
}

// The same kernel for a layer whose input is LayerNorm(X) (the post-LN decoder: every second GEMM of a step): each block
// normalises its 16 rows itself — four rows per wave with layernorm_rows' own arithmetic (ln_wave_stats / ln_apply), X optionally
// being word_table[ids[row]] + pos_row, the step's embedding — into an LDS image the fragments are read from, while the W slabs
// of the whole K quarter are already on their way (LDS-DMA issued first).  Column block 0 also writes the normalised rows out
// (`ln_out`: the residual a later GEMM adds).  Saves a LayerNorm launch (and the embedding's two) per GEMM: 4-5 us each at 25 rows.
struct GemmLN {
    GemmF g;                                                 // g.A unused
    const float* X; int64_t ldx;                             // rows to normalise, or
    const int32_t* ids; const float* table; const float* pos_row;   // X[r] = table[ids[r]] + pos_row
    const float* gamma; const float* beta; float eps;
    float* ln_out; int64_t ldl;
    float* colmax;                                           // stream kernel only: [M, ceil(N / 16)] maxima of the stored 16-column tiles
};

template <int NV, int NT, int DEPTH>
__global__ __launch_bounds__(256) void gemm_f32_m16ln_kernel(GemmLN q) {
    const GemmF& p = q.g;
    constexpr int SLAB = NT * 2048, L = 2 * NT;              // W only: NT x 16 rows x 128 B per slab, two LDS-DMA instructions per column tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A image (16 rows x astride), 4 x DEPTH x SLAB of W, reduction buffer
    const int astride = p.K * 4 + 128;                       // odd multiple of 128 B: row parity picks the bank half, as in the ring
    char* aimg = smem;
    char* ringbase = smem + 16 * astride;
    f32x4 (*red)[NT][64] = reinterpret_cast<f32x4 (*)[NT][64]>(ringbase + 4 * DEPTH * SLAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, slot = lane >> 4;
    // with ln_out, one extra column block (the last) only normalises its rows and writes them out: a store's round trip in a block
    // that also multiplies would be waited for together with its W slabs
    const bool ln_only = q.ln_out && blockIdx.x == gridDim.x - 1;
    const int N0 = ln_only ? 0 : blockIdx.x * 16 * NT, M0 = blockIdx.y * 16;
    const bool odd = slot >> 1;
    char* ring = ringbase + wave * DEPTH * SLAB;
    auto swz = [](int row) { const int pr = (row >> 1) & 7; return pr ^ ((((pr >> 1) ^ (pr >> 2)) & 1) << 2); };
    const char* src[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int trow = 8 * (j & 1) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(trow);
        int gn = N0 + 16 * (j >> 1) + trow; gn = gn < p.N ? gn : p.N - 1;
        src[j] = reinterpret_cast<const char*>(p.W + (int64_t)gn * p.ldw) + 16 * chunk;
    }
    Epi4 ep[NT];                                             // wave 0 stores: its epilogue operands start travelling now
    if (wave == 0 && !ln_only) {
#pragma unroll
        for (int u = 0; u < NT; ++u) ep[u] = epilogue_fetch4(p, M0 + idx, N0 + 16 * u + 4 * slot);
    }
    const int nslab = p.K / FK;
    const int quarter = (nslab + 3) >> 2, first = wave * quarter;
    int cnt = nslab - first; cnt = cnt < 0 ? 0 : (cnt > quarter ? quarter : cnt);
    auto dma = [&](int s, int ringslot) {
        const int64_t koff = (int64_t)(first + (s < cnt ? s : cnt - 1)) * (FK * 4);
        char* dst = ring + ringslot * SLAB;
#pragma unroll
        for (int j = 0; j < L; ++j) glds16(src[j] + koff, dst + j * 1024);
    };
    if (cnt > 0 && !ln_only) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) dma(u, u);
    }
    // LayerNorm of rows M0 + 4 wave .. + 3 by this wave (lane owns float4 number lane + 64 i of a row; K = 256 NV, so every lane of
    // every vector is inside the row).  All loads are unconditional (rows past M re-read row M - 1) and issued before the first
    // reduction: guarded loads end up one memory round trip each.
    {
        const int nv = p.K >> 2;
        f32x4 v[4][NV], gam[NV], bet[NV];
        if (q.ids) {
            f32x4 pe[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) pe[i] = *reinterpret_cast<const f32x4*>(q.pos_row + 4 * (lane + 64 * i));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int m = M0 + 4 * wave + rr; m = m < p.M ? m : p.M - 1;
                const float* tr = q.table + (int64_t)q.ids[m] * p.K;
#pragma unroll
                for (int i = 0; i < NV; ++i) v[rr][i] = *reinterpret_cast<const f32x4*>(tr + 4 * (lane + 64 * i));
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int i = 0; i < NV; ++i) v[rr][i] += pe[i];
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int m = M0 + 4 * wave + rr; m = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int i = 0; i < NV; ++i) v[rr][i] = *reinterpret_cast<const f32x4*>(q.X + (int64_t)m * q.ldx + 4 * (lane + 64 * i));
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            gam[i] = *reinterpret_cast<const f32x4*>(q.gamma + 4 * (lane + 64 * i));
            bet[i] = *reinterpret_cast<const f32x4*>(q.beta + 4 * (lane + 64 * i));
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int rl = 4 * wave + rr;
            float mean, rstd;
            ln_wave_stats<NV>(v[rr], nv, p.K, q.eps, lane, mean, rstd);
            const bool store = ln_only && M0 + rl < p.M;                          // (wave-uniform)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                const f32x4 y = ln_apply(v[rr][i], mean, rstd, gam[i], bet[i]);     // rows past M: a copy of row M - 1, never stored
                if (store) *reinterpret_cast<f32x4*>(q.ln_out + (int64_t)(M0 + rl) * q.ldl + 4 * c) = y;
                *reinterpret_cast<f32x4*>(aimg + rl * astride + (c >> 3) * 128 + (((c & 7) ^ swz(rl)) << 4)) = y;
            }
        }
    }
    if (ln_only) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the W prologue: it had the LayerNorm to land
    __syncthreads();
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = ((4 * (slot & 1) + c) ^ swz(idx)) << 4;
    f32x4 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](int s, int ringslot) {
        const char* wb = ring + ringslot * SLAB + idx * 128;
        const char* ab = aimg + idx * astride + (first + s) * 128;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 aq = *reinterpret_cast<const f32x4*>(ab + fo[c]);
            f32x4 wq[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) wq[u] = *reinterpret_cast<const f32x4*>(wb + u * 2048 + fo[c]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a = odd ? aq[2 * e + 1] : aq[2 * e];
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const float w = odd ? wq[u][2 * e + 1] : wq[u][2 * e];
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, a, acc[u], 0, 0, 0);
                }
            }
        }
    };
    if (cnt > 0) {
        int rs = 0;
        for (int s = 0; s < cnt - DEPTH; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * L) : "memory");
            mm(s, rs);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma(s + DEPTH, rs);
            rs = rs + 1 == DEPTH ? 0 : rs + 1;
        }
        const int base = cnt > DEPTH ? cnt - DEPTH : 0;
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (base + u < cnt) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1 - u) * L) : "memory");
                mm(base + u, rs);
                rs = rs + 1 == DEPTH ? 0 : rs + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) red[wave - 1][u][lane] = acc[u];
    }
    __syncthreads();
    if (wave > 0) return;
    const int m = M0 + idx;
    if (m >= p.M) return;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int n = N0 + 16 * u + 4 * slot;
        if (n < p.N) epilogue_apply4(p, ((acc[u] + red[0][u][lane]) + red[1][u][lane]) + red[2][u][lane], ep[u], m, n);
    }
}

template <int NV, int NT, int DEPTH>
int launch_m16ln(const GemmLN& q, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_f32_m16ln_kernel<NV, NT, DEPTH>;
    constexpr int LDS_MAX = 16 * (NV * 1024 + 128) + 4 * DEPTH * NT * 2048 + 3 * NT * 1024;
    static_assert(LDS_MAX <= 160 * 1024, "does not fit the LDS");
    if (int e = hirest_configure(kern, LDS_MAX, cfg)) return e;
    const int lds = 16 * (q.g.K * 4 + 128) + 4 * DEPTH * NT * 2048 + 3 * NT * 1024;
    hipLaunchKernelGGL(kern, dim3((q.g.N + 16 * NT - 1) / (16 * NT) + (q.ln_out ? 1 : 0), (q.g.M + 15) / 16), dim3(256), lds, s, q);
    return hirest_launch_status();
}

// The LM head (N = 30 522 columns, 94 MB of weights per word): one PERSISTENT block per CU.  It normalises the rows once (four per
// wave, through an LDS image), and every wave then takes the A fragments of ITS K quarter into registers for good (MT x 48 floats,
// already selected for the lane's k slots) — the rows are the same for every column tile.  From there on only W moves: each wave
// streams its quarter of the block's 16-column tiles through a private LDS-DMA ring that never drains between tiles, D0 slabs of
// it in a fixed region that starts filling at launch, D1 more in the space the A image occupied.  With 25 rows that is 8 slabs
// (16 KB) in flight per wave against 2-3 for a kernel that keeps A in LDS, a third of its LDS reads and no per-tile launch,
// first-slab latency and drain.  G groups of four waves (the K quarters) own separate tile sequences; G = MT, so the 4 G waves hold
// the 16 MT rows of the LayerNorm prologue.  At a tile's end a group's quarters meet in LDS and its wave 0 stores while the others
// already multiply the next tile.
template <int NV, int MT, int D0, int D1>
__global__ __launch_bounds__(256 * MT) void gemm_f32_m16ln_stream_kernel(GemmLN q) {
    const GemmF& p = q.g;
    constexpr int G = MT, SLAB = 2048, L = 2, DEPTH = D0 + D1, NS = 2 * NV;   // NS slabs per K quarter (K = 256 NV)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A image (M rows; later D1 slabs per wave), 4 G x D0 slabs, reduction buffers
    const int astride = p.K * 4 + 128;
    char* aimg = smem;
    char* fixed = smem + p.M * astride;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, kq = wave & 3;
    f32x4 (*red)[MT][64] = reinterpret_cast<f32x4 (*)[MT][64]>(fixed + 4 * G * D0 * SLAB) + 3 * grp;
    const int idx = lane & 15, slot = lane >> 4;
    const bool odd = slot >> 1;
    char* ring0 = fixed + wave * D0 * SLAB;                  // ring slots 0 .. D0 - 1
    char* ring1 = aimg + wave * D1 * SLAB;                   // ring slots D0 .. DEPTH - 1 (once the image has been read)
    auto slot_ptr = [&](int rs) { return rs < D0 ? ring0 + rs * SLAB : ring1 + (rs - D0) * SLAB; };
    auto swz = [](int row) { const int pr = (row >> 1) & 7; return pr ^ ((((pr >> 1) ^ (pr >> 2)) & 1) << 2); };
    const int ntile = (p.N + 15) / 16;
    const int tile0 = (int)blockIdx.x * G + grp, tstep = (int)gridDim.x * G;      // this group's tiles: tile0 + i tstep
    const int mine = tile0 < ntile ? (ntile - 1 - tile0) / tstep + 1 : 0;
    const int most = (int)blockIdx.x * G < ntile ? (ntile - 1 - (int)blockIdx.x * G) / tstep + 1 : 0;   // group 0's count = the block's barrier count
    const int first = kq * NS;
    const int items = mine * NS;                             // this wave's stream: (tile i, slab sl), i-major
    // producer side of the stream: the next item to request is slab psl of tile pi, into ring slot prs.  The two row pointers of a
    // lane (LDS-DMA instruction j moves rows 8 j .. 8 j + 7 of the tile) change once per tile; per slab only the k offset moves.
    const int trow0 = lane >> 3, chunk_lane = lane & 7;
    const char* wrow[L];
    auto tile_rows = [&](int i) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int trow = 8 * j + trow0;
            int gn = (tile0 + i * tstep) * 16 + trow; gn = gn < p.N ? gn : p.N - 1;
            wrow[j] = reinterpret_cast<const char*>(p.W + (int64_t)gn * p.ldw) + 16 * (chunk_lane ^ swz(trow)) + (int64_t)first * (FK * 4);
        }
    };
    int psl = 0, pi = 0, prs = 0, issued = 0;
    tile_rows(0);
    auto dma = [&]() {
        char* dst = slot_ptr(prs);
#pragma unroll
        for (int j = 0; j < L; ++j) glds16(wrow[j] + psl * (FK * 4), dst + j * 1024);
        ++issued;
        prs = prs + 1 == DEPTH ? 0 : prs + 1;
        if (++psl == NS) { psl = 0; ++pi; tile_rows(pi); }   // (rows past the last tile are clamped: never requested)
    };
#pragma unroll
    for (int u = 0; u < D0; ++u)
        if (issued < items) dma();
    // LayerNorm of rows 4 wave .. 4 wave + 3 by this wave (see gemm_f32_m16ln_kernel)
    {
        const int nv = p.K >> 2;
        f32x4 gam[NV], bet[NV], v[4][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            gam[i] = *reinterpret_cast<const f32x4*>(q.gamma + 4 * (lane + 64 * i));
            bet[i] = *reinterpret_cast<const f32x4*>(q.beta + 4 * (lane + 64 * i));
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            int m = 4 * wave + rr; m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[rr][i] = *reinterpret_cast<const f32x4*>(q.X + (int64_t)m * q.ldx + 4 * (lane + 64 * i));
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int rl = 4 * wave + rr;
            float mean, rstd;
            ln_wave_stats<NV>(v[rr], nv, p.K, q.eps, lane, mean, rstd);
            if (rl < p.M) {                                  // (wave-uniform)
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane + 64 * i;
                    *reinterpret_cast<f32x4*>(aimg + rl * astride + (c >> 3) * 128 + (((c & 7) ^ swz(rl)) << 4)) =
                        ln_apply(v[rr][i], mean, rstd, gam[i], bet[i]);
                }
            }
        }
    }
    __syncthreads();
    // this wave's A fragments for good: row tile t, slab sl, instruction 2 c + e -> the lane's k slot (rows past M read row M - 1)
    float af[MT][NS][8];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + idx, mr = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 aq = *reinterpret_cast<const f32x4*>(aimg + mr * astride + (first + sl) * 128 + (((4 * (slot & 1) + c) ^ swz(mr)) << 4));
                af[t][sl][2 * c] = odd ? aq[1] : aq[0];
                af[t][sl][2 * c + 1] = odd ? aq[3] : aq[2];
            }
    }
    __syncthreads();                                         // the image is dead: its space becomes ring slots D0 .. DEPTH - 1
#pragma unroll
    for (int u = D0; u < DEPTH; ++u)
        if (issued < items) dma();
    // W fragment of instruction 2 c + e: the lane's k slot is element 2 e + odd of chunk 4 (slot & 1) + c -> two dwords 8 B apart
    // from a per-lane address (one ds_read2_b32, no select)
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = idx * 128 + (((4 * (slot & 1) + c) ^ swz(idx)) << 4) + (odd ? 4 : 0);
    f32x4 acc[MT];
    int it = 0, rs = 0;
    for (int i = 0; i < most; ++i) {
        const bool have = i < mine;                          // (group-uniform; a group without a tile i still meets the barriers)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        Epi4 ep[MT];                                         // the storing wave's epilogue operands travel under the tile's slabs
        if (have && kq == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t) ep[t] = epilogue_fetch4(p, 16 * t + idx, (tile0 + i * tstep) * 16 + 4 * slot);
        }
        if (have) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl, ++it) {
                // slabs it .. min(it + DEPTH, items) - 1 are in flight, in order: the oldest has landed once at most the others are pending
                if (items - it >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const char* wb = slot_ptr(rs);
                float wq[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wp = reinterpret_cast<const float*>(wb + fo[c]);
                    wq[c][0] = wp[0]; wq[c][1] = wp[2];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[c][e], af[t][sl][2 * c + e], acc[t], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot's fragment reads have returned before it is refilled
                rs = rs + 1 == DEPTH ? 0 : rs + 1;
                if (issued < items) dma();
            }
        }
        // (bare barriers: __syncthreads() also waits for vmcnt(0), i.e. empties every wave's W ring at every tile — round 5)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the previous tile's sums have been read
        if (have && kq > 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t) red[kq - 1][t][lane] = acc[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (have && kq == 0) {
            const int tile = tile0 + i * tstep, n = tile * 16 + 4 * slot;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = 16 * t + idx;
                const bool in = n < p.N && m < p.M;
                const f32x4 v = epilogue_apply4(p, ((acc[t] + red[0][t][lane]) + red[1][t][lane]) + red[2][t][lane], ep[t], m, n, in);
                if (q.colmax) {                              // (wave-uniform) the tile's maximum per row: the beam tail's row max without a row scan
                    float mx = in ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : -INFINITY;
                    { float pa = mx, pb = mx; lane_swap16(pa, pb); mx = fmaxf(pa, pb); pa = mx; pb = mx; lane_swap32(pa, pb); mx = fmaxf(pa, pb); }
                    if (slot == 0 && m < p.M) q.colmax[(int64_t)m * ntile + tile] = mx;
                }
            }
        }
    }
}

template <int NV, int MT, int D0, int D1>
int launch_m16ln_stream(const GemmLN& q, hipStream_t s, int lds_max) {
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = gemm_f32_m16ln_stream_kernel<NV, MT, D0, D1>;
    if (int e = hirest_configure(kern, lds_max, cfg, &cus)) return e;
    const int lds = q.g.M * (q.g.K * 4 + 128) + 4 * MT * D0 * 2048 + 3 * MT * MT * 1024;
    if (lds > lds_max || 4 * MT * D1 * 2048 > q.g.M * (q.g.K * 4 + 128)) return HIREST_E_SHAPE;
    const int ntile = (q.g.N + 15) / 16, nblk = (ntile + MT - 1) / MT;
    hipLaunchKernelGGL(kern, dim3(nblk < cus ? nblk : cus), dim3(256 * MT), lds, s, q);
    return hirest_launch_status();
}

// The LM head of a MERGED beam search (33 .. 256 rows: 20 - 32 videos x 3 - 5 beams; round 5).  Too many rows for one block's registers,
// and by now a compute problem, not a weight stream: 160 x 30 528 x 768 is 7.5 GFLOP = 48 us of v_mfma_f32_16x16x4_f32 on 256 CUs,
// while the 94 MB of W take 12 us of HBM.  So: NG row groups of 16 MT rows; a block owns one row group and one of the column-tile
// streams, its four waves = the four K quarters of the shared summation order, each keeping the A fragments of its quarter in
// registers for good (MT x 48 floats, read once from the already-normalised rows) and streaming W through a private LDS-DMA ring that
// never drains: the tile loop has ONE bare barrier per tile (s_waitcnt lgkmcnt(0) + s_barrier — __syncthreads() also waits for vmcnt(0),
// i.e. empties the ring at every tile), the four partial tiles meet in a double-buffered LDS buffer, and the wave that adds them,
// applies the bias (pre-loaded into LDS: no vector load inside the loop, so no compiler-placed vmcnt(0) either) and stores rotates
// with the tile number, so no wave is the one the others wait for.  Fragments of slab s + 1 are read while slab s multiplies.
// The NG blocks that stream the same tiles sit on the same XCD (blockIdx % 8) next to each other and run in step, so W is pulled
// from memory once and served to the other row groups by that XCD's L2.
// With LN, the rows are LayerNorm(X) (X optionally table[ids] + pos_row), as in gemm_f32_m16ln_kernel: each wave first takes the
// statistics of 4 MT of the block's rows with layernorm_rows' own arithmetic (ln_wave_stats; the column-stream-0 blocks also write the
// normalised rows to ln_out), and the A phase then normalises the raw fragments it reads from the ring element by element with
// ln_apply's expression — same bits as hirest_layernorm + the plain product.  A block normalises its rows ONCE for all its column
// tiles, where the 16-row blocks of gemm_f32_m16ln_kernel do it per column tile and run 1450 blocks (three rounds) for a 160 x 2304
// layer.
template <int NS, int MT, int DEPTH, bool LN>
__global__ __launch_bounds__(256) void gemm_f32_rows_stream_kernel(GemmLN q, int ng, int max_mine) {
    static_assert(NS % 2 == 0, "the fragment double buffer alternates by slab parity");
    const GemmF& p = q.g;
    constexpr int SLAB = 2048, L = 2, GA = MT * NS;          // the first GA slabs of a wave's stream are its A tiles, then W
    constexpr int K = NS * 4 * FK;
    // 4 x DEPTH slabs | red[2][4][MT][64] f32x4 | LN: gamma, beta, pos [K] each, (mean, rstd) of the 16 MT rows | bias of this block's tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, kq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, slot = lane >> 4;
    const bool odd = slot >> 1;
    char* ring = smem + kq * DEPTH * SLAB;
    f32x4 (*red)[4][MT][64] = reinterpret_cast<f32x4 (*)[4][MT][64]>(smem + 4 * DEPTH * SLAB);
    float* lnp = reinterpret_cast<float*>(smem + 4 * DEPTH * SLAB + 2 * 4 * MT * 1024);
    float* stats = lnp + 3 * K;
    float* bias_lds = LN ? stats + 2 * 16 * MT : lnp;
    auto swz = [](int row) { const int pr = (row >> 1) & 7; return pr ^ ((((pr >> 1) ^ (pr >> 2)) & 1) << 2); };
    // block -> (XCD, row group, column stream): the ng row groups of a stream are consecutive slots of one XCD
    const int xcd = (int)blockIdx.x & 7, xslot = (int)blockIdx.x >> 3;
    const int rg = xslot % ng, cs = xslot / ng;
    const int ncs = (((int)gridDim.x >> 3) / ng) * 8;       // launcher: gridDim.x = 8 ng k
    const int stream = cs * 8 + xcd;
    const int ntile = (p.N + 15) / 16;
    const int mine = stream < ntile ? (ntile - 1 - stream) / ncs + 1 : 0;      // tiles stream, stream + ncs, ...
    if (mine == 0) return;                                   // (block-uniform)
    const int M0 = rg * 16 * MT;
    const int first = kq * NS;
    const int items = mine * NS, total = GA + items;         // this wave's stream: A (tile t, slab sl), then W (tile i, slab sl)
    // The bias of this block's tiles -> LDS (tile i at floats 16 i .. + 15), by LDS-DMA from wave 0 BEFORE its ring starts: the
    // oldest vector-memory operation of the wave, so every later counted wait implies it, and the only vector loads of the loop
    // stay the ring's (a plain load anywhere in the loop makes the compiler wait for vmcnt(0) = empty the ring).
    if (kq == 0 && p.bias) {
        for (int r = 0; r < max_mine; r += 16) {
            const int i = r + (lane >> 2);
            int n = (stream + (i < mine ? i : mine - 1) * ncs) * 16 + 4 * (lane & 3);
            n = n + 4 <= p.N ? n : p.N - 4;
            glds16(p.bias + n, bias_lds + 16 * r);
        }
    }
    const int trow0 = lane >> 3, chunk_lane = lane & 7;
    const char* wrow[L];
    auto a_rows = [&](int t) {                               // A tile t as a "column tile": its 16 rows x this wave's K quarter
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int trow = 8 * j + trow0;
            int gm = M0 + 16 * t + trow; gm = gm < p.M ? gm : p.M - 1;
            const float* row = !LN ? p.A + (int64_t)gm * p.lda : (q.ids ? q.table + (int64_t)q.ids[gm] * K : q.X + (int64_t)gm * q.ldx);
            wrow[j] = reinterpret_cast<const char*>(row) + 16 * (chunk_lane ^ swz(trow)) + (int64_t)first * (FK * 4);
        }
    };
    auto w_rows = [&](int i) {
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int trow = 8 * j + trow0;
            int gn = (stream + i * ncs) * 16 + trow; gn = gn < p.N ? gn : p.N - 1;
            wrow[j] = reinterpret_cast<const char*>(p.W + (int64_t)gn * p.ldw) + 16 * (chunk_lane ^ swz(trow)) + (int64_t)first * (FK * 4);
        }
    };
    int psl = 0, pi = -MT, prs = 0, issued = 0;              // producer: slab psl of tile pi (pi < 0: A tile pi + MT)
    a_rows(0);
    auto dma = [&]() {
        char* dst = ring + prs * SLAB;
#pragma unroll
        for (int j = 0; j < L; ++j) glds16(wrow[j] + psl * (FK * 4), dst + j * 1024);
        ++issued;
        prs = prs + 1 == DEPTH ? 0 : prs + 1;
        if (++psl == NS) {
            psl = 0; ++pi;
            if (pi < 0) a_rows(pi + MT); else w_rows(pi);    // (rows past the last tile are clamped: never requested)
        }
    };
#pragma unroll 1
    for (int u = 0; u < DEPTH; ++u)
        if (issued < total) dma();
    if (LN) {
        // gamma | beta | pos_row -> LDS in natural order (the A phase reads the lane's k positions from there)
        constexpr int NV = K / 256;
        if (tid < K / 4) {
            reinterpret_cast<f32x4*>(lnp)[tid] = *reinterpret_cast<const f32x4*>(q.gamma + 4 * tid);
            reinterpret_cast<f32x4*>(lnp + K)[tid] = *reinterpret_cast<const f32x4*>(q.beta + 4 * tid);
            reinterpret_cast<f32x4*>(lnp + 2 * K)[tid] = q.ids ? *reinterpret_cast<const f32x4*>(q.pos_row + 4 * tid) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // statistics of rows M0 + 4 (4 j + kq) + rr, j < MT, by this wave (lane owns float4 number lane + 64 i of a row), four rows at
        // a time; all loads unconditional (rows past M re-read row M - 1) and issued before the first reduction
        const bool writes = q.ln_out && cs == 0 && xcd == 0;
        f32x4 gam[NV], bet[NV], pe[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            gam[i] = *reinterpret_cast<const f32x4*>(q.gamma + 4 * (lane + 64 * i));
            bet[i] = *reinterpret_cast<const f32x4*>(q.beta + 4 * (lane + 64 * i));
            pe[i] = q.ids ? *reinterpret_cast<const f32x4*>(q.pos_row + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int j = 0; j < MT; ++j) {
            f32x4 v[4][NV];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int m = M0 + 4 * (4 * j + kq) + rr; m = m < p.M ? m : p.M - 1;
                const float* row = q.ids ? q.table + (int64_t)q.ids[m] * K : q.X + (int64_t)m * q.ldx;
#pragma unroll
                for (int i = 0; i < NV; ++i) v[rr][i] = *reinterpret_cast<const f32x4*>(row + 4 * (lane + 64 * i));
            }
            if (q.ids) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int i = 0; i < NV; ++i) v[rr][i] += pe[i];
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int rl = 4 * (4 * j + kq) + rr;
                float mean, rstd;
                ln_wave_stats<NV>(v[rr], K >> 2, K, q.eps, lane, mean, rstd);
                if (lane == 0) { stats[2 * rl] = mean; stats[2 * rl + 1] = rstd; }
                if (writes && M0 + rl < p.M) {               // (wave-uniform)
#pragma unroll
                    for (int i = 0; i < NV; ++i)
                        *reinterpret_cast<f32x4*>(q.ln_out + (int64_t)(M0 + rl) * q.ldl + 4 * (lane + 64 * i)) = ln_apply(v[rr][i], mean, rstd, gam[i], bet[i]);
                }
            }
        }
        __syncthreads();                                     // statistics and parameters complete (the only full barrier: it also lands the ring's first slabs)
    }
    // fragment of instruction 2 c + e: element 2 e + odd of chunk 4 (slot & 1) + c -> two dwords 8 B apart (one ds_read2_b32)
    int fo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fo[c] = idx * 128 + (((4 * (slot & 1) + c) ^ swz(idx)) << 4) + (odd ? 4 : 0);
    auto read_frags = [&](int rs, float (&w)[4][2]) {
        const char* wb = ring + rs * SLAB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* wp = reinterpret_cast<const float*>(wb + fo[c]);
            w[c][0] = wp[0]; w[c][1] = wp[2];
        }
    };
    // slabs g .. issued - 1 are in flight, in order: slab g has landed once at most DEPTH - 1 slabs are pending (or, near the end of
    // the stream, once nothing is)
    auto landed = [&](int g) {
        if (total - g >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // this wave's A fragments for good: row tile t, slab sl, instruction 2 c + e -> the lane's k slot
    float af[MT][NS][4][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        float mean = 0.f, rstd = 0.f;
        if (LN) { mean = stats[2 * (16 * t + idx)]; rstd = stats[2 * (16 * t + idx) + 1]; }
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int g = t * NS + sl;                       // (compile-time after unrolling)
            landed(g);
            read_frags(g % DEPTH, af[t][sl]);
            if (LN) {
                const int k0 = (first + sl) * FK + 16 * (slot & 1) + (odd ? 1 : 0);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = k0 + 4 * c + 2 * e;
                        float v = af[t][sl][c][e];
                        if (q.ids) v += lnp[2 * K + k];      // (uniform) token + position embedding
                        af[t][sl][c][e] = __builtin_fmaf((v - mean) * rstd, lnp[k], lnp[K + k]);       // ln_apply's expression
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot's reads have returned before it is refilled
            if (issued < total) dma();
        }
    }
    float wq[2][4][2];
    landed(GA);
    read_frags(GA % DEPTH, wq[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (issued < total) dma();
    f32x4 acc[MT];
    int it = 0, rs = (GA + 1) % DEPTH;                       // rs: ring slot of W slab it + 1
#pragma unroll 1
    for (int i = 0; i < mine; ++i) {
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < NS; ++sl, ++it) {
            const bool more = it + 1 < items;
            if (more) {                                      // fragments of slab it + 1 are read while slab it multiplies
                landed(GA + it + 1);
                read_frags(rs, wq[(sl + 1) & 1]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[sl & 1][c][e], af[t][sl][c][e], acc[t], 0, 0, 0);
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rs = rs + 1 == DEPTH ? 0 : rs + 1;
                if (issued < total) dma();
            }
        }
        // The four K quarters of tile i meet in red[i & 1], and EVERY wave reduces its share of the row tiles — t with (t + i) % 4 = kq —
        // in the shared order and stores them.  (One reducing wave per tile, rotating or not, puts its whole epilogue on the critical
        // path: the reducer of tile i + 1 waits for the quarter of the wave that was busy storing tile i — profiles/r05/lm_head_rows_forms.txt.)
        const int buf = i & 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) red[buf][kq][t][lane] = acc[t];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const int tile = stream + i * ncs, n = tile * 16 + 4 * slot;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 bias4 = p.bias ? *reinterpret_cast<const f32x4*>(bias_lds + 16 * i + 4 * slot) : zero;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (((t + i) & 3) != kq) continue;           // (wave-uniform)
                const int m = M0 + 16 * t + idx;
                const bool in = n < p.N && m < p.M;
                Epi4 ep;                                     // (no residual / periodic operand here: the launcher sends those elsewhere)
                ep.bias = bias4; ep.periodic = zero; ep.resid = zero;
                const f32x4 sum = ((red[buf][0][t][lane] + red[buf][1][t][lane]) + red[buf][2][t][lane]) + red[buf][3][t][lane];
                const f32x4 v = epilogue_apply4(p, sum, ep, m, n, in);
                if (q.colmax) {                              // (uniform) the tile's maximum per row: the beam tail's row max without a row scan
                    float mx = in ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : -INFINITY;
                    { float pa = mx, pb = mx; lane_swap16(pa, pb); mx = fmaxf(pa, pb); pa = mx; pb = mx; lane_swap32(pa, pb); mx = fmaxf(pa, pb); }
                    if (slot == 0 && m < p.M) q.colmax[(int64_t)m * ntile + tile] = mx;
                }
            }
        }
    }
}

template <int NS, int MT, int DEPTH, bool LN>
int launch_rows_stream(const GemmLN& q, int ng, int k, hipStream_t s) {
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = gemm_f32_rows_stream_kernel<NS, MT, DEPTH, LN>;
    constexpr int K = NS * 4 * FK;
    constexpr int FIXED = 4 * DEPTH * 2048 + 2 * 4 * MT * 1024 + (LN ? 3 * K * 4 + 2 * 16 * MT * 4 : 0);
    static_assert(FIXED + 4096 <= 160 * 1024, "does not fit the LDS");
    if (int e = hirest_configure(kern, 160 * 1024, cfg, &cus)) return e;
    if (q.g.resid || q.g.periodic) return HIREST_E_SHAPE;
    const int ntile = (q.g.N + 15) / 16, ncs = 8 * k;
    const int max_mine = (ntile + ncs - 1) / ncs;
    const int lds = FIXED + ((max_mine + 15) / 16) * 16 * 64;      // the bias pre-load writes whole 1-KiB LDS-DMA pieces (16 tiles each)
    if (lds > 160 * 1024) return HIREST_E_SHAPE;
    hipLaunchKernelGGL(kern, dim3(8 * ng * k), dim3(256), lds, s, q, ng, max_mine);
    return hirest_launch_status();
}

// M rows as ng groups of mt 16-row tiles (mt <= 5: MT x 48 fragment registers per wave) x 8 k column streams, one block per CU: the
// split with the shortest block — tiles per stream x (MFMA time of a tile + its reduction) + the prologue (statistics, A phase)
// hirest_gemm_f32_rows_ln_mode (A/B): 0 one block per CU, ring depth 12, 1 - 3 row tiles per wave; 1 / 2: one row tile per wave, ring depth 4,
// one / two blocks per CU.  Default 2: the layer products of a merged search (160 x 2304 / 3072 x 768) are a few microseconds of matrix
// time behind a LayerNorm prologue, and a second block per CU covers one block's prologue with the other's MFMAs (B = 32, beam 5:
// 1735 -> 1800 captions/s; mode 1: 1755)
static std::atomic<int> g_rows_ln_mode{2};
extern "C" int hirest_gemm_f32_rows_ln_mode(int32_t mode) {
    if (mode < 0 || mode > 2) return HIREST_E_BADARG;
    g_rows_ln_mode = mode;
    return 0;
}
template <bool LN>
static int rows_stream(const GemmLN& q, hipStream_t s) {
    static HirestDevCfg cfg;
    int cus = 0;
    if (int e = hirest_configure(gemm_f32_rows_stream_kernel<6, 1, 12, LN>, 160 * 1024, cfg, &cus)) return e;
    const int per_xcd = cus / 8 < 1 ? 1 : cus / 8;          // blockIdx % 8 = XCD
    const int tiles = (q.g.M + 15) / 16, ntile = (q.g.N + 15) / 16;
    if (LN && g_rows_ln_mode) {
        int k = (per_xcd * g_rows_ln_mode) / tiles; if (k < 1) k = 1;
        if (8 * k > ntile) k = (ntile + 7) / 8;
        return launch_rows_stream<6, 1, 4, LN>(q, tiles, k, s);
    }
    int best_mt = 0, best_ng = 0, best_k = 0;
    int64_t best = 0;
    for (int mt = 1; mt <= (LN ? 3 : 5); ++mt) {             // (the LayerNorm form of 4 / 5 row tiles does not fit the registers)
        const int ng = (tiles + mt - 1) / mt;
        int k = per_xcd / ng; if (k < 1) k = 1;
        if (8 * k > ntile) k = (ntile + 7) / 8;
        const int per_stream = (ntile + 8 * k - 1) / (8 * k);
        const int64_t cost = (int64_t)per_stream * (mt * 6 * 8 * 42 + 800) + (int64_t)mt * (LN ? 2500 : 1500);
        if (best_mt == 0 || cost < best) { best = cost; best_mt = mt; best_ng = ng; best_k = k; }
    }
    switch (best_mt) {
        case 1: return launch_rows_stream<6, 1, 12, LN>(q, best_ng, best_k, s);
        case 2: return launch_rows_stream<6, 2, 12, LN>(q, best_ng, best_k, s);
        case 3: return launch_rows_stream<6, 3, 12, LN>(q, best_ng, best_k, s);
        case 4: return launch_rows_stream<6, 4, 12, false>(q, best_ng, best_k, s);
        default: return launch_rows_stream<6, 5, 12, false>(q, best_ng, best_k, s);
    }
}
static int rows_stream_plain(const GemmF& p, float* colmax, hipStream_t s) {
    if (p.resid || p.periodic) return HIREST_E_SHAPE;
    GemmLN q{p, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, colmax};
    return rows_stream<false>(q, s);
}

// ---------------------------------------------------------------------------------------------
// fp32 flash attention, head dim 64, full (unmasked) attention over T keys with the reference's uniform
// additive constant: s = fl(fl(q.k * scale) + add_const) (module_visual.py:164-176 with the all-zeros mask
// of modeling.py:208 -> -10000 on every score, SURVEY hazard H3), softmax, @ v.
// One wave per 32 queries; keys in tiles of 32 staged through LDS by the 4 waves of a block.
//   S^T = K.Q^T   (lane = query, registers = keys)  -> softmax statistics are lane-local + one shuffle
//   O^T = V^T.P^T  P is reused in place as the B operand: k-step r pairs key kap(r,0) (lanes < 32) with
//                  kap(r,1) = kap(r,0)+4 (lanes >= 32), which is exactly what each half-wave holds in reg r.
// ---------------------------------------------------------------------------------------------
// (shared by the two attention kernels, the fused multiply-adds written out so that no kernel contracts differently from the
// other: the scale and the additive masks in one rounding, as this kernel has always computed them)
__device__ __forceinline__ float attn_score(float st, float scale, float addc, bool valid) {
    const float sv = __builtin_fmaf(st, scale, addc);
    return valid ? sv : -3.0e38f;
}
__device__ __forceinline__ float attn_lsum(float lrun, float alpha, float psum) { return __builtin_fmaf(lrun, alpha, psum); }

// DHP = head dim padded to a multiple of 32 (64, or 96 for EVA-CLIP's 88-wide heads: the fp32 reference-precision tower);
// dh = the real head dim (the packed layouts are addressed with it; padded dims are zeros and their outputs are not stored).
// NW = waves per block (32 queries each): 4, or 1 for short sequences (the sentence encoder's 4 .. 40-token sentences left three of
// four waves without a query, staging and synchronising for nothing).
// (min 2 waves per SIMD for the multi-wave blocks: left alone the compiler gave the 96-wide, three-wave instantiation — EVA's 88-wide heads
//  at 257 tokens — 214 VGPRs + 48 AGPRs = one wave per SIMD, i.e. one 3-wave block per CU with a SIMD idle: 40 % matrix-pipe duty.)
template <int DHP, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 1 : (DHP > 64 ? 3 : 3)) void attention_f32_kernel(const float* __restrict__ qp, int64_t ldq, const float* __restrict__ kp,
                                                           const float* __restrict__ vp, int64_t ldkv, float* __restrict__ out,
                                                           int Tq, int T, int H, int dh, float scale, float add_const,
                                                           float causal_penalty, const int32_t* __restrict__ seq_off) {
    constexpr int ALD = DHP + 1, NO = DHP / 32;
    __shared__ float Ks[32 * ALD];
    __shared__ float Vs[32 * ALD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int QB = 32 * NW, NT = 64 * NW;                // queries per block, threads per block
    const int qblocks = (Tq + QB - 1) / QB;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
    const int b = bh / H, h = bh - b * H;
    const int D = H * dh;
    int64_t qrow0 = (int64_t)b * Tq, krow0 = (int64_t)b * T;
    if (seq_off) {   // packed ragged self-attention: sequence b = rows seq_off[b] .. seq_off[b+1]; Tq was only the longest one
        qrow0 = krow0 = seq_off[b];
        Tq = T = seq_off[b + 1] - seq_off[b];
        if (qb * QB >= Tq) return;                       // (block-uniform)
    }
    const float* qbase = qp + qrow0 * ldq + h * dh;
    const float* kbase = kp + krow0 * ldkv + h * dh;
    const float* vbase = vp + krow0 * ldkv + h * dh;
    const int q = qb * QB + wave * 32 + l31;
    const bool qvalid = q < Tq;
    // Q^T fragments: B operand [k = d][j = query]: lane holds Q[q][2s + half] for s = 0..DHP/2-1
    // (every load unconditional on a clamped index, the zero chosen afterwards: a guarded load is a memory round trip of its own)
    float qf[DHP / 2];
    const float* qrow = qbase + (int64_t)(qvalid ? q : Tq - 1) * ldq;
#pragma unroll
    for (int s = 0; s < DHP / 2; ++s) {
        const int d = 2 * s + half;
        const float v = qrow[d < dh ? d : dh - 1];
        qf[s] = (qvalid && d < dh) ? v : 0.f;
    }
    f32x16 o[NO];    // O^T rows d = 32 j .. 32 j + 31, column = query
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // K / V tile staging through registers, one tile ahead (the next tile's rows travel while this one is multiplied: a synchronous
    // load per 32-key tile was a memory round trip in front of every tile, 10 of them at T = 300)
    constexpr int NLD = (32 * (DHP / 4) + NT - 1) / NT;       // float4 per thread and operand: 2 (dh 64) or 3 (96) with four waves
    f32x4 kreg[NLD], vreg[NLD];
    auto fetch_kv = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            int i = tid + NT * u; i = i < 32 * (DHP / 4) ? i : 32 * (DHP / 4) - 1;
            const int kr = i / (DHP / 4), c = (i - kr * (DHP / 4)) * 4;
            const int key = k0 + kr < T ? k0 + kr : T - 1;
            const int cc = c < dh ? c : dh - 4;                  // (dh % 4 == 0: a 4-vector is inside or outside the head as a whole)
            kreg[u] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * ldkv + cc);
            vreg[u] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * ldkv + cc);
        }
    };
    fetch_kv(0);
    for (int k0 = 0; k0 < T; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + NT * u;
            if (i < 32 * (DHP / 4)) {
                const int kr = i / (DHP / 4), c = (i - kr * (DHP / 4)) * 4;
                const f32x4 kv = c >= dh ? zero4 : kreg[u], vv = c >= dh ? zero4 : vreg[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) { Ks[kr * ALD + c + e] = kv[e]; Vs[kr * ALD + c + e] = vv[e]; }
            }
        }
        __syncthreads();
        if (k0 + 32 < T) fetch_kv(k0 + 32);
        f32x16 st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
        // (operand reads in groups of eight, fenced: left to itself the scheduler hoists all DHP / 2 LDS reads of the chain — and the
        //  48 of the P V loop below — in front of the first MFMA, 40-odd live registers that cost the kernel a wave per SIMD)
#pragma unroll
        for (int g8 = 0; g8 < DHP / 2; g8 += 8) {
            float kf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kf[i] = Ks[l31 * ALD + 2 * (g8 + i) + half];   // A operand: K[key = l31][d = 2s + half]
#pragma unroll
            for (int i = 0; i < 8; ++i) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[i], qf[g8 + i], st, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float tmax = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sv = attn_score(st[r], scale, key > q ? add_const + causal_penalty : add_const, key < T);
            st[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        { float pa = tmax, pb = tmax; lane_swap32(pa, pb); tmax = fmaxf(pa, pb); }   // the other half-wave's keys (common.h: wave_max_x)
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float pz = __expf(st[r] - mnew); st[r] = pz; psum += pz; }
        { float pa = psum, pb = psum; lane_swap32(pa, pb); psum = pa + pb; }
        lrun = attn_lsum(lrun, alpha, psum);
        mrun = mnew;
#pragma unroll
        for (int j = 0; j < NO; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // k-step r: keys kap(r,0) | kap(r,1); A operand V^T[d = l31 (+32 j)][key]
            const int kr = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int j = 0; j < NO; ++j)
                o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kr * ALD + 32 * j + l31], st[r], o[j], 0, 0, 0);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!qvalid) return;
    const float inv = 1.0f / lrun;
    float* orow = out + (qrow0 + q) * D + h * dh;
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // O^T rows (reg&3) + 8*(reg>>2) + 4*half = d
            const int d = 32 * j + 8 * g + 4 * half;
            if (d < dh)
                *reinterpret_cast<f32x4*>(orow + d) = f32x4{o[j][4 * g] * inv, o[j][4 * g + 1] * inv, o[j][4 * g + 2] * inv, o[j][4 * g + 3] * inv};
        }
}

// One query per (row, head) — a beam-search step: the kernel above spends a 32 x 32 MFMA tile (two dependent chains of 32 + 32
// v_mfma_f32_32x32x2_f32 = 4096 cycles per key tile) on a single query column, after a separate kernel has re-gathered every
// beam's K / V history by parent row.  Here one wave owns (row, head) and does the same arithmetic in the same order with vector
// FMAs (an fp32 MFMA is its k-ordered fma chain: tools/probes/fma_order_probe.hip): scores with lane = key (chain over d = 0..63),
// the 32-key tile's max / exp / ordered sums as in the kernel above, P.V with lane = d (chain over the keys in the MFMA's order:
// k-step r pairs key kap(r) = (r & 3) + 8 (r >> 2) with kap(r) + 4).  The history is read in place through the parent row (and the
// newest key from the packed q | k | v rows); the gathered + appended K / V rows are written out on the way, which is all that
// kv_gather_append_kernel did.
struct AttnDec {
    const float* q; int64_t ldq;
    const float* k_hist; const float* v_hist; int64_t ld_hist; const int32_t* parent; int t_hist;
    const float* k_new; const float* v_new; int64_t ld_new;
    float* k_out; float* v_out;
    float* out;
    int H; float scale, add_const, causal_penalty;
};

__global__ __launch_bounds__(64) void attention_f32_decode_kernel(AttnDec p) {
    constexpr int KLD = 68;
    __shared__ __attribute__((aligned(16))) float Ks[64 * KLD];
    __shared__ __attribute__((aligned(16))) float Qs[64];
    const int lane = threadIdx.x;
    const int r = blockIdx.x / p.H, h = blockIdx.x - r * p.H;
    const int D = p.H * 64, T = p.t_hist + (p.k_new ? 1 : 0);
    const int64_t src = p.parent ? p.parent[r] : r;
    Qs[lane] = p.q[(int64_t)r * p.ldq + h * 64 + lane];
    float mrun = -3.0e38f, lrun = 0.f, o = 0.f;
    for (int k0 = 0; k0 < T; k0 += 64) {
        __syncthreads();
        // K rows k0 .. k0 + 63 -> LDS (4 keys x 16 float4 per pass), V column d = lane of the same keys -> registers.  Every load is
        // unconditional (keys past T re-read key T - 1) and issued before the first use; then both are written out as the row's new
        // history (a store interleaved with its load costs the load's round trip each time).
        f32x4 kreg[16];
        float vv[64];
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int key = k0 + ps * 4 + (lane >> 4), kk = key < T ? key : T - 1;
            const float* kp = kk < p.t_hist ? p.k_hist + (src * p.t_hist + kk) * p.ld_hist : p.k_new + (int64_t)r * p.ld_new;
            kreg[ps] = *reinterpret_cast<const f32x4*>(kp + h * 64 + 4 * (lane & 15));
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int key = k0 + j, kk = key < T ? key : T - 1;
            const float* vp = kk < p.t_hist ? p.v_hist + (src * p.t_hist + kk) * p.ld_hist : p.v_new + (int64_t)r * p.ld_new;
            vv[j] = vp[h * 64 + lane];
        }
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) *reinterpret_cast<f32x4*>(Ks + (ps * 4 + (lane >> 4)) * KLD + 4 * (lane & 15)) = kreg[ps];
        if (p.k_out) {
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int key = k0 + ps * 4 + (lane >> 4);
                if (key < T) *reinterpret_cast<f32x4*>(p.k_out + ((int64_t)r * T + key) * D + h * 64 + 4 * (lane & 15)) = kreg[ps];
            }
#pragma unroll
            for (int j = 0; j < 64; ++j)
                if (k0 + j < T) p.v_out[((int64_t)r * T + k0 + j) * D + h * 64 + lane] = vv[j];
        }
        __syncthreads();
        float st = 0.f;                                       // lane = key k0 + lane: q . k, d = 0 .. 63 in order
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 k4 = *reinterpret_cast<const f32x4*>(Ks + lane * KLD + 4 * c);
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(Qs + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_fmaf(k4[e], q4[e], st);
        }
        const int key = k0 + lane;
        const float sv = attn_score(st, p.scale, key > 0 ? p.add_const + p.causal_penalty : p.add_const, key < T);
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            if (k0 + 32 * tile >= T) break;                   // (uniform)
            // max of the tile's 32 scores: rows of 16 by DPP, the two rows by a swap (lanes 0-31 = tile 0, 32-63 = tile 1)
            float t = sv;
            t = fmaxf(t, lane_dpp<0x128>(t)); t = fmaxf(t, lane_dpp<0x124>(t)); t = fmaxf(t, lane_dpp<0x4E>(t)); t = fmaxf(t, lane_dpp<0xB1>(t));
            { float a = t, b = t; lane_swap16(a, b); t = fmaxf(a, b); }
            const float tmax = fmaxf(-3.0e38f, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 32 * tile)));
            const float mnew = fmaxf(mrun, tmax);
            const float alpha = __expf(mrun - mnew);
            const float pz = __expf(sv - mnew);               // (meaningful in this tile's lanes)
            float pk[32];                                     // the tile's probabilities, wave-uniform
#pragma unroll
            for (int j = 0; j < 32; ++j) pk[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz), 32 * tile + j));
            float ps0 = 0.f, ps1 = 0.f;                       // the two half-waves' sums of the kernel above, each in its register order
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { const int kap = (rr & 3) + 8 * (rr >> 2); ps0 += pk[kap]; ps1 += pk[kap + 4]; }
            lrun = attn_lsum(lrun, alpha, ps0 + ps1);
            mrun = mnew;
            o *= alpha;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int kap = (rr & 3) + 8 * (rr >> 2);
                o = __builtin_fmaf(vv[32 * tile + kap], pk[kap], o);
                o = __builtin_fmaf(vv[32 * tile + kap + 4], pk[kap + 4], o);
            }
        }
    }
    const float inv = 1.0f / lrun;
    p.out[(int64_t)r * D + h * 64 + lane] = o * inv;
}

// head width (a multiple of 4) -> the narrowest instantiation that holds it: 32 (MiniLM's 32-wide heads: the sentence encoder used to
// zero-pad them to 64, i.e. twice the attention and qkv / output-projection work), 64 (the joint model), 96 (EVA-CLIP's 88).  Padded
// dims are zeros that join the sums last or not at all, so a head gives the same bits in every instantiation that holds it.
// Waves (32 queries each) per block: one when no sequence is longer than 64 — at most two one-wave blocks then replace a four-wave
// block that would be at least half empty; otherwise four, or three when that wastes fewer waves (257 queries, the fp32 EVA-CLIP
// tower: 3 x 3 waves instead of 2 four-wave blocks + one that serves a single query).  A query's arithmetic does not depend on it.
template <int NW, class... Args>
void launch_attention_f32_nw(int dh, dim3 grid, hipStream_t s, Args... args) {
    if (dh <= 32) hipLaunchKernelGGL((attention_f32_kernel<32, NW>), grid, dim3(64 * NW), 0, s, args...);
    else if (dh <= 64) hipLaunchKernelGGL((attention_f32_kernel<64, NW>), grid, dim3(64 * NW), 0, s, args...);
    else hipLaunchKernelGGL((attention_f32_kernel<96, NW>), grid, dim3(64 * NW), 0, s, args...);
}
template <class... Args>
int launch_attention_f32(int dh, int64_t BH, int Tq_max, hipStream_t s, Args... args) {
    if (dh <= 0 || dh % 4 != 0 || dh > 96) return HIREST_E_SHAPE;
    const int waves = (Tq_max + 31) / 32;
    const int nw = Tq_max <= 64 ? 1 : ((waves + 2) / 3 * 3 < (waves + 3) / 4 * 4 ? 3 : 4);
    const dim3 grid((unsigned)(BH * ((waves + nw - 1) / nw)));
    if (nw == 1) launch_attention_f32_nw<1>(dh, grid, s, args...);
    else if (nw == 3) launch_attention_f32_nw<3>(dh, grid, s, args...);
    else launch_attention_f32_nw<4>(dh, grid, s, args...);
    return hirest_launch_status();
}

// base[b,t,:] = v[b,t,:] * tn[b,:] + asr[b,t,:] + temporal[b,t,:]     (loop-invariant part of modeling.py:167-195)
__global__ __launch_bounds__(256) void joint_base_kernel(const float* __restrict__ v, const float* __restrict__ tproj,
                                                        const float* __restrict__ asr, const float* __restrict__ temporal,
                                                        float* __restrict__ base, int B, int T, int E) {
    __shared__ float red[4];
    __shared__ float inv_norm;
    // one block per (b, chunk of rows); text L2 norm recomputed per block (E = 512: trivial)
    const int b = blockIdx.y;
    const float* tp = tproj + (int64_t)b * E;
    float ss = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) ss += tp[e] * tp[e];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) inv_norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    const float nrm = inv_norm;
    const int nv = E >> 2;
    const int rows_per_block = 16;
    const int t0 = blockIdx.x * rows_per_block;
    for (int i = threadIdx.x; i < rows_per_block * nv; i += 256) {
        const int t = t0 + i / nv, c = (i % nv) * 4;
        if (t >= T) break;
        const int64_t off = ((int64_t)b * T + t) * E + c;
        const f32x4 vv = *reinterpret_cast<const f32x4*>(v + off);
        f32x4 tn = *reinterpret_cast<const f32x4*>(tp + c);
        tn[0] /= nrm; tn[1] /= nrm; tn[2] /= nrm; tn[3] /= nrm;
        f32x4 r = vv * tn;
        r += *reinterpret_cast<const f32x4*>(asr + off);
        r += *reinterpret_cast<const f32x4*>(temporal + off);
        *reinterpret_cast<f32x4*>(base + off) = r;
    }
}

__device__ __forceinline__ float time_grid_value(int t, int n) {
    if (t >= n) return 0.f;
    // torch.linspace(0, 1, n): step = 1/(n-1); values i*step for the first half, 1-(n-1-i)*step for the second
    float lin;
    if (n == 1) lin = 0.f;
    else {
        const float step = 1.0f / (float)(n - 1);
        lin = t < n / 2 ? (float)t * step : 1.0f - (float)(n - 1 - t) * step;
    }
    return (lin - 0.5f) * 2.0f;
}
// the grid itself, [B, T]: the weight of temporal_embed.0.weight's gradient column sum (the training step must not fetch n_valid
// to the host to build it: that is a device synchronisation in the middle of the backward)
__global__ __launch_bounds__(256) void joint_time_grid_kernel(const int32_t* __restrict__ n_valid, float* __restrict__ grid, int B, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * T) grid[i] = time_grid_value(i % T, n_valid[i / T]);
}
// temporal input: tin[b,t,:] = tanh(time(b,t) * w1 + b1), time = (linspace(0,1,n_b)[t]-0.5)*2 for t < n_b else 0
// (modeling.py:176-195; linspace(0,1,1) = [0]).  One thread per 4 channels.
__global__ __launch_bounds__(256) void joint_time_kernel(const int32_t* __restrict__ n_valid, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, float* __restrict__ tin, int B, int T, int E) {
    const int nv = E >> 2;
    const int64_t total = (int64_t)B * T * nv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % nv) * 4;
        const int64_t row = idx / nv;
        const int t = (int)(row % T), b = (int)(row / T);
        const float tm = time_grid_value(t, n_valid[b]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tanhf(tm * w1[c + e] + b1[c + e]);
        *reinterpret_cast<f32x4*>(tin + row * E + c) = o;
    }
}

// f[b,t,:] = base[b,t,:] (+ boundary_embed[bmask[b,t]]) + mask_embed[mmask[b,t]]    (modeling.py:171-173,197-198)
__global__ __launch_bounds__(256) void joint_mask_add_kernel(const float* __restrict__ base, const int32_t* __restrict__ mmask,
                                                            const int32_t* __restrict__ bmask, const float* __restrict__ mask_embed,
                                                            const float* __restrict__ boundary_embed, float* __restrict__ f,
                                                            int64_t rows, int E) {
    const int nv = E >> 2;
    const int64_t total = rows * nv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % nv) * 4;
        const int64_t row = idx / nv;
        f32x4 r = *reinterpret_cast<const f32x4*>(base + row * E + c);
        if (bmask) r += *reinterpret_cast<const f32x4*>(boundary_embed + (int64_t)bmask[row] * E + c);
        r += *reinterpret_cast<const f32x4*>(mask_embed + (int64_t)mmask[row] * E + c);
        *reinterpret_cast<f32x4*>(f + row * E + c) = r;
    }
}

// up to 3 Linear(D,1) heads: logits[h][row] = <x[row], w_h> + b_h  (modeling.py:218-219,319); one wave per row
__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ x, int64_t rows, int D, int nheads,
                                                   const float* __restrict__ w0, const float* __restrict__ w1,
                                                   const float* __restrict__ w2, const float* __restrict__ bias3,
                                                   float* __restrict__ logits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 a = *reinterpret_cast<const f32x4*>(w0 + c);
        s0 += xv[0] * a[0] + xv[1] * a[1] + xv[2] * a[2] + xv[3] * a[3];
        if (nheads > 1) { const f32x4 bq = *reinterpret_cast<const f32x4*>(w1 + c); s1 += xv[0] * bq[0] + xv[1] * bq[1] + xv[2] * bq[2] + xv[3] * bq[3]; }
        if (nheads > 2) { const f32x4 cq = *reinterpret_cast<const f32x4*>(w2 + c); s2 += xv[0] * cq[0] + xv[1] * cq[1] + xv[2] * cq[2] + xv[3] * cq[3]; }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        logits[row] = s0 + bias3[0];
        if (nheads > 1) logits[rows + row] = s1 + bias3[1];
        if (nheads > 2) logits[2 * rows + row] = s2 + bias3[2];
    }
}

// per-sample masked argmax: out[b] = argmax_t (mask[b,t] ? logits[b,t] : fill)   (modeling.py:294-298), first maximum
__global__ __launch_bounds__(256) void masked_argmax_kernel(const float* __restrict__ logits, const int32_t* __restrict__ mask,
                                                           float fill, int T, int32_t* __restrict__ out) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int t = tid; t < T; t += 256) {
        const float v = mask[(int64_t)b * T + t] ? logits[(int64_t)b * T + t] : fill;
        if (v > best || (v == best && t < bi)) { best = v; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (rv[w] > best || (rv[w] == best && ri[w] < bi)) { best = rv[w]; bi = ri[w]; }
        out[b] = bi;
    }
}

// One iteration of the segmentation loop for every sample (modeling.py:393-433), entirely on device:
// masked softmax over T, argmax, threshold walk (Python-float = double arithmetic, like scores.tolist()),
// zero moment_mask[l..r], set boundary_mask[l] and [r], append [l,r] to the sample's step list.
__global__ __launch_bounds__(256) void segmentation_step_kernel(const float* __restrict__ logits, int32_t* __restrict__ mmask,
                                                               int32_t* __restrict__ bmask, int T, double threshold,
                                                               int32_t* __restrict__ steps, int32_t* __restrict__ nsteps,
                                                               int max_steps, float* __restrict__ probs_out) {
    extern __shared__ float pr[];     // [T] probabilities
    __shared__ float red[4];
    __shared__ int redi[4];
    __shared__ float s_max, s_sum;
    __shared__ int s_arg;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* mm = mmask + (int64_t)b * T;
    int32_t* bm = bmask + (int64_t)b * T;
    const float NEG = -3.4028234663852886e38f;   // -finfo(float32).max
    float mx = -INFINITY;
    for (int t = tid; t < T; t += 256) {
        const float v = mm[t] ? logits[(int64_t)b * T + t] : NEG;
        pr[t] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) s_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = s_max;
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) { const float e = expf(pr[t] - mx); pr[t] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) s_sum = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float tot = s_sum;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int t = tid; t < T; t += 256) {
        const float pz = pr[t] / tot;
        pr[t] = pz;
        if (probs_out) probs_out[(int64_t)b * T + t] = pz;
        if (pz > best || (pz == best && t < bi)) { best = pz; bi = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
        s_arg = bi;
    }
    __syncthreads();
    if (tid == 0) {
        const int max_idx = s_arg;
        const double max_score = (double)pr[max_idx];
        if (!(max_score < 0.00001)) {
            int left = max_idx, right = max_idx;
            while (((double)pr[left] / max_score) > threshold) { if (left == 0) break; --left; }
            while (((double)pr[right] / max_score) > threshold) { if (right == T - 1) break; ++right; }
            if (!(left == 0 || right == 0)) {
                for (int t = left; t <= right; ++t) mm[t] = 0;
                bm[left] = 1; bm[right] = 1;
                const int n = nsteps[b];
                if (n < max_steps) { steps[((int64_t)b * max_steps + n) * 2] = left; steps[((int64_t)b * max_steps + n) * 2 + 1] = right; nsteps[b] = n + 1; }
            }
        }
    }
}

inline int grid1d(int64_t total, int cap = 4096) { int64_t g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

}  // namespace

extern "C" int hirest_gemm_f32_rows_preferred(int32_t M);
static std::atomic<int> g_f32_ring{0};         // hirest_gemm_f32_ring_mode (A/B): 0 automatic, 1 off (the register-prefetch kernel), 2 always the ring
extern "C" int hirest_gemm_f32_ring_mode(int32_t mode) {
    if (mode < 0 || mode > 2) return HIREST_E_BADARG;
    g_f32_ring = mode;
    return 0;
}
static std::atomic<int> g_f32_kernel{0};       // hirest_gemm_f32_select_kernel: A/B and tests
// 0 automatic (16-column kernel for M <= 256 when K % 32 == 0 [N < 8192 above 32 rows], else the split-K 32x32 kernel for M <= 256),
// 1 always the 64x64 kernel, 2 automatic without the 16-column kernel
extern "C" int hirest_gemm_f32_select_kernel(int32_t which) {
    if (which < 0 || which > 2) return HIREST_E_BADARG;
    g_f32_kernel = which;
    return 0;
}

extern "C" int hirest_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                               const float* resid, int64_t ldr, const float* periodic, int32_t period,
                               float* out, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 3) return HIREST_E_BADARG;
    if (K % 16 != 0 || N % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || (periodic && period <= 0)) return HIREST_E_SHAPE;
    GemmF p{A, lda, W, ldw, bias, resid, ldr, periodic, period, out, ldo, M, N, K, act, nullptr};
    // 33 .. 256 rows (the sentence encoder's batches, the captioning task's training rows): the same kernel, 32-row tiles across
    // blockIdx.y — whole-line operand traffic beats the split-K kernel's lane = row loads (ASR encoder 108 -> 117 k sentences/s)
    // (a 3072-deep product is a chain of 192 dependent MFMAs per K quarter: one row tile per wave, i.e. twice the blocks, halves it)
    if (M > 32 && M <= 256 && K % FK == 0 && g_f32_kernel == 0 && N < 8192)
        return K >= 2048 ? launch_m16<1, 1, 4>(p, reinterpret_cast<hipStream_t>(stream)) : launch_m16<2, 1, 6>(p, reinterpret_cast<hipStream_t>(stream));
    // a merged beam search's LM head (60 - 160 rows x 30 522 columns): A fragments in registers, W streamed once per row group — when its row
    // groups pad the rows less than 64-row tiles do (96, 150, 160 rows; at 60, 100, 256 the 64x64 kernel below is faster: hirest_gemm_f32_rows_preferred)
    if (M > 32 && M <= 256 && N >= 8192 && K == 768 && !resid && !periodic && g_f32_kernel == 0 && hirest_gemm_f32_rows_preferred(M))
        return rows_stream_plain(p, nullptr, reinterpret_cast<hipStream_t>(stream));
    // other wide problems of 48 - 256 rows: the 64x64 kernel beats the split-K kernel's lane = row loads from 60 rows on (LM head: 63 vs 97 us at
    // 96 rows, 93 vs 150 at 160)
    if (M >= 48 && M <= 256 && N >= 8192 && g_f32_kernel == 0) {
        hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
        return hirest_launch_status();
    }
    if (M <= 32 && K % FK == 0 && g_f32_kernel == 0) {
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        // one row tile per block for the decoder's layers (2 x N / 16 blocks share the operand traffic; 8 slabs in flight per wave,
        // 3 for the 2304- / 3072-wide ones so that two blocks fit a CU: +3 % per word); the LM head without a LayerNorm prologue
        // (hirest_gemm_f32_ln has its own): both row tiles and two column tiles per wave, two slabs in flight (best of 2 / 3 / 4)
        if (N < 2048) return launch_m16<1, 1, 8>(p, s);
        if (N < 8192) return launch_m16<1, 1, 3>(p, s);
        return M <= 16 ? launch_m16<1, 2, 6>(p, s) : launch_m16<2, 2, 2>(p, s);
    }
    if (M <= 256 && g_f32_kernel != 1) {
        hipLaunchKernelGGL(gemm_f32_skinny_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
        return hirest_launch_status();
    }
    // row-major operands in whole 32-deep slabs, dozens of tiles per CU (the fp32 towers' layers): the LDS-DMA ring form of the same tile (same
    // bits), 4 slots and two blocks per CU: 105 -> 111 TFLOP/s at 65 792 rows.  Below ~64 tiles per CU the register-prefetch form with its four
    // blocks per CU is as fast or faster (9600 x 3072 x 768: 103 vs 100), and a deeper ring with one block per CU loses everywhere — a lone wave per
    // SIMD pays for its own DMA issue and fragment latency (profiles/r05/gemm_f32_ring_ab.txt)
    if (g_f32_kernel == 0 && g_f32_ring != 1 && K % FK == 0 && K >= 2 * FK) {
        int cus = 0; static HirestDevCfg cfg;
        if (int e = hirest_configure(gemm_f32_ring_kernel<4>, 4 * 16384, cfg, &cus)) return e;
        const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
        if (g_f32_ring == 2 || tiles >= 64 * (int64_t)cus) return launch_ring<4>(p, reinterpret_cast<hipStream_t>(stream));
    }
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return hirest_launch_status();
}

// Split form of the 64x64 kernel for problems with few tiles (see gemm_f32_kernel): taken for 256 < M, at most 512 tiles, K >= 1024
// (same box, B = 5, T = 300: segmentation 330 -> 371 videos/s, retrieval 5400 -> 5800, training step 5.85 -> 5.55 ms; thresholds of
// 512 .. 2048 for K and 400 .. 1200 tiles measure the same: the gain is the K = 3072 layer)
static bool splits(int M, int N, int K) {
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    // (round 5 lowered the threshold to K = 512 — 1500 x 768 x 768, 288 tiles: 38.4 -> 33.6 us, ~1 % of an inference batch — and that is what took
    //  the training step from 3.07 to 3.45 ms between the round-4 and round-5 bench lines: its dX / dW products at K = 768 each gained a reduce
    //  launch on a launch-bound step.  Same box, same build, round 6: 3.23 / 3.57 ms at 512 against 3.05 / 3.25 at 1024
    //  (profiles/r06/train_bisect.txt).  Back to 1024; HIREST_F32_SPLIT_MIN_K = A/B of the threshold.)
    static const int min_k = [] { const char* e = getenv("HIREST_F32_SPLIT_MIN_K"); return e ? atoi(e) : 1024; }();
    return g_f32_kernel == 0 && M > 256 && tiles <= 512 && K >= min_k;
}
extern "C" size_t hirest_gemm_f32_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    return (M > 0 && N > 0 && K > 0 && splits(M, N, K)) ? (size_t)4 * M * N * 4 : 0;
}
extern "C" int hirest_gemm_f32_ws(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* resid,
                                  int64_t ldr, const float* periodic, int32_t period, float* out, int64_t ldo, int32_t M, int32_t N,
                                  int32_t K, int32_t act, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 3) return HIREST_E_BADARG;
    if (K % 16 != 0 || N % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || (periodic && period <= 0)) return HIREST_E_SHAPE;
    const size_t need = hirest_gemm_f32_workspace_bytes(M, N, K);
    if (need == 0 || !workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
        return hirest_gemm_f32(A, lda, W, ldw, bias, resid, ldr, periodic, period, out, ldo, M, N, K, act, stream);
    GemmF p{A, lda, W, ldw, bias, resid, ldr, periodic, period, out, ldo, M, N, K, act, static_cast<float*>(workspace)};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3((N + 63) / 64, (M + 63) / 64, 4), dim3(256), 0, s, p);
    const int64_t n = (int64_t)M * (N / 4);
    hipLaunchKernelGGL(gemm_f32_quarters_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    return hirest_launch_status();
}

static thread_local float* g_ln_colmax = nullptr;       // set by hirest_gemm_f32_ln_colmax around its call of hirest_gemm_f32_ln
extern "C" int hirest_gemm_f32_ln(const float* X, int64_t ldx, const int32_t* ids, const float* table, const float* pos_row,
                                  const float* gamma, const float* beta, float eps, float* ln_out, int64_t ldl, const float* W,
                                  int64_t ldw, const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M,
                                  int32_t N, int32_t K, int32_t act, void* stream) {
    if ((!X && !(ids && table && pos_row)) || !gamma || !beta || !W || !out || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 3) return HIREST_E_BADARG;
    // the 16-row blocks of the layer form tile any number of rows across blockIdx.y (a merged beam search: 60 - 160 rows per word);
    // the LM-head stream keeps all its rows in one block's registers: 32 at most
    if ((M > 256 && !(K == 768 && !resid)) || K % 256 != 0 || K > 1024 || N % 4 != 0 || ldw % 4 != 0 || (X && ldx % 4 != 0) || (ln_out && ldl % 4 != 0)) return HIREST_E_SHAPE;
    // (narrow layers stay with the 16-row blocks below: 10 vs 12 us at 160 x 768; above 256 rows this is the only form)
    if (M > 32 && K == 768 && !resid && (N >= 2048 || M > 256)) {   // a merged search's rows: row groups x column streams, rows normalised once per block
        GemmLN qs{GemmF{nullptr, 0, W, ldw, bias, nullptr, 0, nullptr, 0, out, ldo, M, N, K, act, nullptr}, X, ldx, ids, table, pos_row, gamma, beta, eps,
                  ln_out, ldl, g_ln_colmax};
        return rows_stream<true>(qs, reinterpret_cast<hipStream_t>(stream));
    }
    if (M > 32 && N >= 8192) return HIREST_E_SHAPE;
    GemmLN q{GemmF{nullptr, 0, W, ldw, bias, resid, ldr, nullptr, 0, out, ldo, M, N, K, act, nullptr}, X, ldx, ids, table, pos_row, gamma, beta, eps,
             ln_out, ldl, g_ln_colmax};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (N >= 8192 && K == 768 && !ids && !ln_out) {          // the LM head: persistent blocks, rows normalised once per CU
        // ring depth: D0 slabs per wave in the fixed region + D1 in the space of the A image (M rows x 3200 B shared by 4 MT waves)
        constexpr int MAXLDS = 160 * 1024;
        if (M <= 2) return launch_m16ln_stream<3, 1, 8, 0>(q, s, MAXLDS);
        if (M <= 5) return launch_m16ln_stream<3, 1, 8, 1>(q, s, MAXLDS);
        if (M <= 7) return launch_m16ln_stream<3, 1, 8, 2>(q, s, MAXLDS);
        if (M <= 10) return launch_m16ln_stream<3, 1, 8, 3>(q, s, MAXLDS);
        if (M <= 12) return launch_m16ln_stream<3, 1, 8, 4>(q, s, MAXLDS);
        if (M <= 16) return launch_m16ln_stream<3, 1, 8, 5>(q, s, MAXLDS);
        if (M <= 20) return launch_m16ln_stream<3, 2, 4, 3>(q, s, MAXLDS);
        if (M <= 26) return launch_m16ln_stream<3, 2, 4, 4>(q, s, MAXLDS);
        return launch_m16ln_stream<3, 2, 2, 5>(q, s, MAXLDS);
    }
    if (N >= 8192) return HIREST_E_SHAPE;
    switch (K / 256) {
        case 1: return launch_m16ln<1, 1, 2>(q, s);
        case 2: return launch_m16ln<2, 1, 2>(q, s);
        case 3: return launch_m16ln<3, 1, 2>(q, s);   // two W slabs in flight per wave (66 KB of LDS, two blocks per CU: best of 2 / 3 / 6;
                                                      // two column tiles per wave for the wide layers measured no better)
        default: return launch_m16ln<4, 1, 2>(q, s);
    }
}

// The LM-head form of hirest_gemm_f32_ln (N >= 8192, K = 768, no ids, no ln_out) that also reports, per row, the maximum of every
// 16-column tile it stores: colmax [M, ceil(N / 16)].
extern "C" int hirest_gemm_f32_ln_colmax(const float* X, int64_t ldx, const float* gamma, const float* beta, float eps, const float* W,
                                         int64_t ldw, const float* bias, float* out, int64_t ldo, float* colmax, int32_t M, int32_t N,
                                         int32_t K, void* stream) {
    if (!colmax) return HIREST_E_BADARG;
    if (N < 8192 || K != 768) return HIREST_E_SHAPE;
    g_ln_colmax = colmax;
    const int e = hirest_gemm_f32_ln(X, ldx, nullptr, nullptr, nullptr, gamma, beta, eps, nullptr, 0, W, ldw, bias, nullptr, 0, out, ldo, M, N, K, 0, stream);
    g_ln_colmax = nullptr;
    return e;
}

// 1 when the row-group streaming kernel pads M rows to fewer rows (groups of 16 x 1..5) than 64-row tiles do: measured on the LM head, it then
// beats the 64x64 kernel (96 rows: 55 vs 62 us, 160: 80 vs 94) and loses otherwise (60: 44 vs 37, 100: 71 vs 62, 256: 129 vs 125)
extern "C" int hirest_gemm_f32_rows_preferred(int32_t M) {
    if (M <= 32) return 0;
    const int tiles = (M + 15) / 16;
    int best = 1 << 30;
    for (int mt = 3; mt <= 5; ++mt) { const int padded = ((tiles + mt - 1) / mt) * mt; best = padded < best ? padded : best; }   // (the splits its chooser takes for wide products)
    return best * 16 < ((M + 63) / 64) * 64 ? 1 : 0;
}

// out = A @ W^T + bias for the rows of a merged beam search (K = 768, N >= 16) through the row-group streaming kernel, which
// also reports colmax[M, ceil(N / 16)]: per row, the maximum of each 16-column tile of `out` (NULL: not wanted).
extern "C" int hirest_gemm_f32_rows_colmax(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                                           float* colmax, int32_t M, int32_t N, int32_t K, void* stream) {
    if (!A || !W || !out || M <= 0 || N <= 0) return HIREST_E_BADARG;
    if (K != 768 || M > 1280 || N % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || ldo % 4 != 0) return HIREST_E_SHAPE;
    GemmF p{A, lda, W, ldw, bias, nullptr, 0, nullptr, 0, out, ldo, M, N, K, 0, nullptr};
    return rows_stream_plain(p, colmax, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int hirest_gemm_f32_layouts(const float* A, int64_t lda, int32_t a_kmajor, const float* W, int64_t ldw, int32_t w_kmajor,
                                       const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M, int32_t N,
                                       int32_t K, int32_t act, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 3) return HIREST_E_BADARG;
    if (N % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0 || (a_kmajor ? (M % 4 != 0 || M < 4) : K % 4 != 0) || (w_kmajor ? N < 4 : K % 4 != 0)) return HIREST_E_SHAPE;
    GemmF p{A, lda, W, ldw, bias, resid, ldr, nullptr, 0, out, ldo, M, N, K, act, nullptr};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // the 64x64 kernel for every size (the few-row kernels read row-major operands only); split form as in hirest_gemm_f32_ws
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    const bool split = tiles <= 512 && K >= 1024 && workspace && workspace_bytes >= (size_t)4 * M * N * 4 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    if (split) p.ws = static_cast<float*>(workspace);
    const dim3 grid((N + 63) / 64, (M + 63) / 64, split ? 4 : 1), blk(256);
    if (a_kmajor && w_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, blk, 0, s, p);
    else if (a_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, blk, 0, s, p);
    else if (w_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, blk, 0, s, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, blk, 0, s, p);
    if (split) {
        const int64_t n = (int64_t)M * (N / 4);
        hipLaunchKernelGGL(gemm_f32_quarters_kernel, dim3((unsigned)((n + 255) / 256)), blk, 0, s, p);
    }
    return hirest_launch_status();
}
extern "C" size_t hirest_gemm_f32_layouts_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    return (tiles <= 512 && K >= 1024) ? (size_t)4 * M * N * 4 : 0;
}

extern "C" int hirest_attention_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t H, int32_t dh,
                                    float scale, float add_const, void* stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || H <= 0) return HIREST_E_BADARG;
    const int64_t ld = 3 * (int64_t)H * dh;
    return launch_attention_f32(dh, (int64_t)B * H, T, reinterpret_cast<hipStream_t>(stream), qkv, ld, qkv + H * dh, qkv + 2 * H * dh, ld,
                                out, (int)T, (int)T, (int)H, (int)dh, scale, add_const, 0.f, (const int32_t*)nullptr);
}

extern "C" int hirest_attention_f32_varlen(const float* qkv, float* out, const int32_t* seq_off, int32_t B, int32_t max_len, int32_t H,
                                           int32_t dh, float scale, float add_const, void* stream) {
    if (!qkv || !out || !seq_off || B <= 0 || max_len <= 0 || H <= 0) return HIREST_E_BADARG;
    const int64_t ld = 3 * (int64_t)H * dh;
    return launch_attention_f32(dh, (int64_t)B * H, max_len, reinterpret_cast<hipStream_t>(stream), qkv, ld, qkv + H * dh, qkv + 2 * H * dh, ld,
                                out, (int)max_len, (int)max_len, (int)H, (int)dh, scale, add_const, 0.f, seq_off);
}

extern "C" int hirest_attention_f32_qkv(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out,
                                        int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, float add_const,
                                        float causal_penalty, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0) return HIREST_E_BADARG;
    if (ldq % 4 != 0 || ldkv % 4 != 0) return HIREST_E_SHAPE;
    return launch_attention_f32(dh, (int64_t)B * H, Tq, reinterpret_cast<hipStream_t>(stream), q, ldq, k, v, ldkv, out, (int)Tq, (int)Tk,
                                (int)H, (int)dh, scale, add_const, causal_penalty, (const int32_t*)nullptr);
}

extern "C" int hirest_attention_f32_decode(const float* q, int64_t ldq, const float* k_hist, const float* v_hist, int64_t ld_hist,
                                           const int32_t* parent, int32_t t_hist, const float* k_new, const float* v_new, int64_t ld_new,
                                           float* k_out, float* v_out, float* out, int32_t R, int32_t H, float scale, float add_const,
                                           float causal_penalty, void* stream) {
    if (!q || !out || R <= 0 || H <= 0 || t_hist < 0 || (t_hist > 0 && (!k_hist || !v_hist)) || (!k_new != !v_new) || (!k_out != !v_out)) return HIREST_E_BADARG;
    if (t_hist == 0 && !k_new) return HIREST_E_BADARG;
    if (ldq % 4 != 0 || ld_hist % 4 != 0 || ld_new % 4 != 0) return HIREST_E_SHAPE;
    AttnDec p{q, ldq, k_hist, v_hist, ld_hist, parent, t_hist, k_new, v_new, ld_new, k_out, v_out, out, H, scale, add_const, causal_penalty};
    hipLaunchKernelGGL(attention_f32_decode_kernel, dim3(R * H), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p);
    return hirest_launch_status();
}

// out[r][v] = x[r][v] - logsumexp(x[r]) + row_add[r]   (log_softmax of train.py:563-564 fused with the beam score add of beam.py:76)
// One 1024-thread block per row (beam search: 5-25 rows of 30 522 logits, so the row itself has to supply the parallelism):
// 16-byte loads where the row allows them, three passes over a row that stays in the L2.
__global__ __launch_bounds__(1024) void log_softmax_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ row_add,
                                                          float* __restrict__ out, int64_t ldo, int V, int vec) {
    __shared__ float red[16];
    __shared__ float bc;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (int64_t)r * ldx;
    float* o = out + (int64_t)r * ldo;
    const int V4 = vec ? V >> 2 : 0;                      // 4-wide part (all of the row when vec), scalar tail otherwise
    float mx = -INFINITY;
    for (int i = tid; i < V4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * i);
        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    for (int i = 4 * V4 + tid; i < V; i += 1024) mx = fmaxf(mx, xr[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
        for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
        bc = m;
    }
    __syncthreads();
    mx = bc;
    float s = 0.f;
    for (int i = tid; i < V4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * i);
        s += (expf(v[0] - mx) + expf(v[1] - mx)) + (expf(v[2] - mx) + expf(v[3] - mx));
    }
    for (int i = 4 * V4 + tid; i < V; i += 1024) s += expf(xr[i] - mx);
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];          // fixed order
        bc = logf(t);
    }
    __syncthreads();
    const float lse = bc, add = row_add ? row_add[r] : 0.f;
    for (int i = tid; i < V4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * i);
        *reinterpret_cast<f32x4*>(o + 4 * i) = f32x4{((v[0] - mx) - lse) + add, ((v[1] - mx) - lse) + add, ((v[2] - mx) - lse) + add,
                                                     ((v[3] - mx) - lse) + add};
    }
    for (int i = 4 * V4 + tid; i < V; i += 1024) o[i] = ((xr[i] - mx) - lse) + add;
}

extern "C" int hirest_log_softmax_f32(const float* x, int64_t ldx, const float* row_add, float* out, int64_t ldo, int32_t rows,
                                      int32_t V, void* stream) {
    if (!x || !out || rows <= 0 || V <= 0) return HIREST_E_BADARG;
    const int vec = (V % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, row_add, out, ldo, V, vec);
    return hirest_launch_status();
}

extern "C" int hirest_joint_time_features(const int32_t* n_valid, const float* w1, const float* b1, float* tin, int32_t B,
                                          int32_t T, int32_t E, void* stream) {
    if (!n_valid || !w1 || !b1 || !tin || B <= 0 || T <= 0 || E % 4 != 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(joint_time_kernel, dim3(grid1d((int64_t)B * T * (E / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       n_valid, w1, b1, tin, B, T, E);
    return hirest_launch_status();
}

extern "C" int hirest_joint_time_grid_f32(const int32_t* n_valid, int32_t B, int32_t T, float* grid, void* stream) {
    if (!n_valid || !grid || B <= 0 || T <= 0 || (int64_t)B * T > INT32_MAX) return HIREST_E_BADARG;
    hipLaunchKernelGGL(joint_time_grid_kernel, dim3((B * T + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n_valid, grid,
                       B, T);
    return hirest_launch_status();
}

extern "C" int hirest_joint_base(const float* v, const float* text_proj, const float* asr, const float* temporal, float* base,
                                 int32_t B, int32_t T, int32_t E, void* stream) {
    if (!v || !text_proj || !asr || !temporal || !base || B <= 0 || T <= 0 || E % 4 != 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(joint_base_kernel, dim3((T + 15) / 16, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), v, text_proj,
                       asr, temporal, base, B, T, E);
    return hirest_launch_status();
}

extern "C" int hirest_joint_mask_add(const float* base, const int32_t* moment_mask, const int32_t* boundary_mask,
                                     const float* mask_embed, const float* boundary_embed, float* f, int64_t rows, int32_t E,
                                     void* stream) {
    if (!base || !moment_mask || !mask_embed || !f || rows <= 0 || E % 4 != 0) return HIREST_E_BADARG;
    if (boundary_mask && !boundary_embed) return HIREST_E_BADARG;
    hipLaunchKernelGGL(joint_mask_add_kernel, dim3(grid1d(rows * (E / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), base,
                       moment_mask, boundary_mask, mask_embed, boundary_embed, f, rows, E);
    return hirest_launch_status();
}

extern "C" int hirest_linear_heads(const float* x, int64_t rows, int32_t D, int32_t nheads, const float* w0, const float* w1,
                                   const float* w2, const float* bias3, float* logits, void* stream) {
    if (!x || !w0 || !bias3 || !logits || rows <= 0 || nheads < 1 || nheads > 3 || D % 4 != 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(heads_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, rows, D,
                       nheads, w0, w1 ? w1 : w0, w2 ? w2 : w0, bias3, logits);
    return hirest_launch_status();
}

extern "C" int hirest_masked_argmax(const float* logits, const int32_t* mask, float fill, int32_t B, int32_t T, int32_t* out,
                                    void* stream) {
    if (!logits || !mask || !out || B <= 0 || T <= 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(masked_argmax_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, mask, fill, T, out);
    return hirest_launch_status();
}

extern "C" int hirest_segmentation_step(const float* logits, int32_t* moment_mask, int32_t* boundary_mask, int32_t B, int32_t T,
                                        double threshold, int32_t* steps, int32_t* nsteps, int32_t max_steps, float* probs_out,
                                        void* stream) {
    if (!logits || !moment_mask || !boundary_mask || !steps || !nsteps || B <= 0 || T <= 0 || T > 16384) return HIREST_E_BADARG;
    hipLaunchKernelGGL(segmentation_step_kernel, dim3(B), dim3(256), T * sizeof(float), reinterpret_cast<hipStream_t>(stream), logits,
                       moment_mask, boundary_mask, T, threshold, steps, nsteps, max_steps, probs_out);
    return hirest_launch_status();
}
