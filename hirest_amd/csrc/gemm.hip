// bf16 x bf16 -> fp32 GEMM on MFMA with fused epilogues, "TN" form: C[m][n] = sum_k A[m][k] W[n][k]
// (W is an nn.Linear weight as stored).  This is 97 % of the EVA-CLIP-g FLOPs
// (vit_model.py:56-62,124-127,148): QKV, proj, fc1(+GELU), fc2(+residual), patch-embed, head.
//
// Kernel "t128": 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16.  Operands are staged HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B), double buffered,
// one barrier per K-step.  The LDS image is [row][8 x 16-B chunk] with chunk' = chunk ^ ((row>>1)&7):
// the XOR is applied on the per-lane GLOBAL source address (LDS-DMA writes lane-linear) and
// again on the ds_read_b128 fragment address, which makes every 16-lane ds_read_b128 group hit
// 16 distinct 16-B slots of the 256-B bank row (conflict-free).
//
// MFMA operands are swapped (a = W fragment, b = A fragment) so each lane ends up owning 4
// CONSECUTIVE output columns of one output row (D rows = n, D col = m): epilogue loads/stores
// are 8-B (bf16) / 16-B (f32) vectors and bias is a 16-B load.
#include <cstdio>
#include <atomic>
#include "gemm_shared.h"

namespace {


constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmP& p, int m, int n, f32x4 v) {
    // v = 4 consecutive columns n..n+3 of row m
    if (p.bias) {
        f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v += b;
    }
    if constexpr (EPI == HIREST_EPI_BIAS_BF16 || EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_BIAS_QGELU_BF16) {
        if constexpr (EPI == HIREST_EPI_BIAS_GELU_BF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
        }
        if constexpr (EPI == HIREST_EPI_BIAS_QGELU_BF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = quick_gelu(v[i]);
        }
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if constexpr (EPI == HIREST_EPI_BIAS_RESID_F32) {
        float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n;
        f32x4 r = *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = r + v;
    } else if constexpr (EPI == HIREST_EPI_BIAS_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = v;
    } else if constexpr (EPI == HIREST_EPI_BIAS_GELU_SPLIT2) {
        // GELU (erf form, the epilogue_p form's arithmetic) stored as bf16 hi | lo in the split operand format: the 4 columns lie in one 32-column
        // block (n % 4 == 0), whose hi half starts at 64 (n / 32) and whose lo half 32 elements further
#if HIREST_X3_GELU_POLY
        const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
        const f32x4 gv = {g0[0], g0[1], g1[0], g1[1]};
#else
        f32x4 gv;
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752440f));
#endif
        bf16x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { hi[i] = (bf16_t)gv[i]; lo[i] = (bf16_t)(gv[i] - (float)hi[i]); }
        bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (int64_t)m * p.ldo + (n >> 5) * 64 + (n & 31);
        *reinterpret_cast<bf16x4*>(o) = hi;
        *reinterpret_cast<bf16x4*>(o + 32) = lo;
    } else {  // HIREST_EPI_PATCH_POS_F32
        const int b = m / p.P, pp = m - b * p.P;
        const int64_t orow = (int64_t)b * (p.P + 1) + 1 + pp;
        f32x4 ps = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(1 + pp) * p.N + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v + ps;
    }
}

// (The split-operand small-problem kernel is its own kernel further down: gemm_t128x3.)
template <int EPI>
__device__ __forceinline__ void t128_body(const GemmP& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- block -> tile, XCD-aware: XCD x (= blockIdx % 8 in practice) owns a contiguous range of
    // M-panels; inside it tiles are walked GROUP_M panels at a time, n outer / m inner, so the ~64
    // tiles resident on one XCD form a near-square patch that shares A and W panels in its L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int p_lo = xcd * p.ppx;
    int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
    if (np <= 0 || j >= np * p.nbn) return;
    const int grp = j / (GROUP_M * p.nbn);
    const int r = j - grp * GROUP_M * p.nbn;
    int gcount = np - grp * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
    const int nt = r / gcount, mt = p_lo + grp * GROUP_M + (r - nt * gcount);
    const int M0 = mt * BM, N0 = nt * BN;

    // ---- staging addresses: piece q of this wave = rows (wave*4+q)*8 + lane/8, one 16-B chunk per lane
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1;
        a_src[q] = p.A + (int64_t)gm * p.lda + chunk * 8;
        w_src[q] = p.W + (int64_t)gn * p.ldw + chunk * 8;
    }
    auto stage = [&](int s, int kt) {
        char* As = smem + s * STAGE_BYTES;
        char* Ws = As + BM * BK * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(a_src[q] + (int64_t)kt * BK, As + (wave * 4 + q) * 1024);
            glds16(w_src[q] + (int64_t)kt * BK, Ws + (wave * 4 + q) * 1024);
        }
    };

    // ---- fragment read offsets
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fsw = (frow >> 1) & 7, khalf = lane >> 5;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + khalf) ^ fsw) << 4;
    const int a_row_off = (wm * 64 + frow) * (BK * 2);
    const int w_row_off = (wn * 64 + frow) * (BK * 2);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* As = smem + (kt & 1) * STAGE_BYTES;
        const char* Ws = As + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(As + a_row_off + i * 32 * BK * 2 + koff[kk]);
                wf[i] = *reinterpret_cast<const bf16x8*>(Ws + w_row_off + i * 32 * BK * 2 + koff[kk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[jn], af[i], acc[i][jn], 0, 0, 0);
        }
    }

    // ---- epilogue: D row (n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), D col (m) = lane&31
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = M0 + wm * 64 + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = N0 + wn * 64 + jn * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= p.N) continue;
                f32x4 v = {acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]};
                epilogue_store<EPI>(p, m, n, v);
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_t128(GemmP p) { t128_body<EPI>(p); }

// Split-operand ("bf16x3") kernel for problems of a few hundred 128 x 128 tiles at most — the joint model's training GEMMs (1 500 rows,
// 768 - 3 072 wide, K = 768 - 3 072): one workgroup per tile and CU, so what matters is how fast ONE tile goes.  Eight waves = the 2 x 2
// wave grid of t128 twice: group kg = wave >> 2 multiplies the kg-th 16-deep chunk of every 64-column step ([hi | lo] of 32 real k: chunk 0
// = k 0..15, chunk 1 = k 16..31, each with its hi and lo columns), i.e. the two groups split K inside the step and read the same 32-KiB
// stage.  Two waves per SIMD cover each other's fragment reads; a 4-slot ring keeps three steps of LDS-DMA in flight (counted vmcnt: a
// step is 4 instructions per wave); one barrier per step.  At the end group 1 hands its partial tile to group 0 through the (drained) ring:
// out = (sum over chunk-0 terms) + (sum over chunk-1 terms), a fixed order.
// (Round 4 also had a K-step-blocked operand layout for this kernel, HIREST_GEMM_KBLOCKED, for the training step's split-operand path; both
// were measured slower end to end and removed in round 5: profiles/r04/train_x3_ab.txt.)
// WM = wave rows of a group: 2 -> 128 x 128 tiles (8 waves), 3 -> 192 x 128 tiles (12 waves, 160 KiB of ring) for the shapes whose 128-row
// panels waste a round: the 160 beam rows of a B = 32 beam-5 LM head (2 x 239 tiles = two rounds, 37 % of them padding -> 239 tiles, one round)
// and the joint encoder's 3072-wide layer at 1 500 rows (288 tiles -> 192).  Same arithmetic per output element: the tile height does not enter
// the k order.
template <int EPI, int WM = 2>
__global__ __launch_bounds__(256 * WM) void gemm_t128x3(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 4;
    constexpr int BMX = 64 * WM, NWG = 2 * WM;                    // tile rows; waves per K group
    constexpr int STAGE = (BMX + BN) * BK * 2;                    // 32 / 40 KiB per ring slot
    constexpr int NPIECE = (BMX + BN) / 8;                        // 1-KiB LDS-DMA pieces per step: A first, then W; piece j belongs to wave j / 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = wave / NWG, w4 = wave - kg * NWG;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    int nt, mt, slice = 0;
    if (p.flat) {
        // few row panels (the joint model: 12 at 1 500 rows): the panel split above gives 6 of the 8 XCDs two panels each and the other two nothing
        // (qkv: 36 tiles on the 32 CUs of six XCDs = two rounds, 64 CUs idle).  Flat split instead: the tile list in (column outer, panel inner)
        // order cut into 8 equal contiguous chunks, one per XCD — each XCD streams every A panel and its own few W panels.
        // With p.ksplit = S > 1 an item is (tile, K slice): slice s multiplies steps [s nk / S, (s + 1) nk / S) and stores its raw partial tile
        // to p.part [S][M][N]; splitk_reduce_kernel adds the slices in slice order, then bias and residual (deterministic).
        const int ntile = p.nbm * p.nbn * p.ksplit, per = (ntile + 7) >> 3, item = xcd * per + j;
        if (j >= per || item >= ntile) return;
        const int tile = item / p.ksplit;
        slice = item - tile * p.ksplit;
        nt = tile / p.nbm; mt = tile - nt * p.nbm;
    } else {
        const int p_lo = xcd * p.ppx;
        int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
        if (np <= 0 || j >= np * p.nbn) return;
        const int grp = j / (GROUP_M * p.nbn);
        const int r = j - grp * GROUP_M * p.nbn;
        int gcount = np - grp * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
        nt = r / gcount; mt = p_lo + grp * GROUP_M + (r - nt * gcount);
    }
    const int M0 = mt * BMX, N0 = nt * BN;

    // staging: piece j = 4 wave + q (8 rows x 128 B): the first BMX / 8 pieces are A rows, the rest W rows; waves past the last piece (WM = 3:
    // waves 10, 11) bring nothing — a wave's counted wait covers its own pieces, the barrier everybody's
    const bf16_t* src[4];
    const bool stager = wave * 4 < NPIECE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pj = wave * 4 + q, row = (pj < BMX / 8 ? pj : pj - BMX / 8) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        if (pj < BMX / 8) { int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1; src[q] = p.A + (int64_t)gm * p.lda + chunk * 8; }
        else { int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1; src[q] = p.W + (int64_t)gn * p.ldw + chunk * 8; }
        src[q] += (int64_t)slice * ((p.K / BK) / p.ksplit) * BK;
    }
    static_assert(NPIECE % 4 == 0, "whole waves of four pieces");
    const int64_t kstep = BK;
    auto stage = [&](int slot, int kt) {
        if (!stager) return;
        char* dst = smem + slot * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(src[q] + (int64_t)kt * kstep, dst + (wave * 4 + q) * 1024);
    };
    const int wm = w4 >> 1, wn = w4 & 1;
    const int frow = lane & 31, fsw = (frow >> 1) & 7, khalf = lane >> 5;
    const int koff_hi = ((kg * 2 + khalf) ^ fsw) << 4, koff_lo = (((kg + 2) * 2 + khalf) ^ fsw) << 4;
    const int a_row_off = (wm * 64 + frow) * (BK * 2);
    const int w_row_off = (wn * 64 + frow) * (BK * 2);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;
    const int nk = (p.K / BK) / p.ksplit;
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) if (i < nk) stage(i, i);
    // One barrier per step; both groups read their fragments, then multiply.  (A ping-pong form — two barriers per step, group 1 half a
    // step behind, one group reading while the other multiplies — was measured 20 % SLOWER: with every load, read and MFMA knocked out the
    // loop still costs ~0.2 us per barrier here, which is most of a step)
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // step kt landed everywhere; everyone is done reading step kt - 1
        if (kt + NST - 1 < nk && !(p.dbg & 1)) stage((kt + NST - 1) % NST, kt + NST - 1);
        const char* As = smem + (kt % NST) * STAGE;
        const char* Ws = As + BMX * BK * 2;
        bf16x8 af[2][2], wf[2][2];                             // [tile][0 hi, 1 lo]
        if (!(p.dbg & 8)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][0] = *reinterpret_cast<const bf16x8*>(As + a_row_off + i * 32 * BK * 2 + koff_hi);
                af[i][1] = *reinterpret_cast<const bf16x8*>(As + a_row_off + i * 32 * BK * 2 + koff_lo);
                wf[i][0] = *reinterpret_cast<const bf16x8*>(Ws + w_row_off + i * 32 * BK * 2 + koff_hi);
                wf[i][1] = *reinterpret_cast<const bf16x8*>(Ws + w_row_off + i * 32 * BK * 2 + koff_lo);
            }
        }
        if (!(p.dbg & 4)) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g3 = 0; g3 < 3; ++g3)                      // (W half, A half): (hi, lo), (lo, hi), (hi, hi)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[jn][g3 == 1 ? 1 : 0], af[i][g3 == 0 ? 1 : 0], acc[i][jn], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // ---- group 1 -> group 0 through LDS: [w4][tile][reg][lane] fp32 = 16 KiB per wave of the group
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + w4 * (4 * 16 * 64);
    if (kg == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((i * 2 + jn) * 16 + e) * 64 + lane] = acc[i][jn][e];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] += red[((i * 2 + jn) * 16 + e) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = M0 + wm * 64 + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = N0 + wn * 64 + jn * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= p.N) continue;
                f32x4 v = {acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]};
                if (p.ksplit > 1) *reinterpret_cast<f32x4*>(p.part + ((int64_t)slice * p.M + m) * p.N + n) = v;
                else epilogue_store<EPI>(p, m, n, v);
            }
        }
    }
}


// out[m][n] += (part[0][m][n] + part[1][m][n] + ...) + bias[n]   (the split-K form of HIREST_EPI_BIAS_RESID_F32; N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias, float* __restrict__ out,
                                                           int64_t ldo, int M, int N) {
    const int nq = N >> 2;
    const int64_t total = (int64_t)M * nq, plane = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / nq;
        const int n = (int)(i - m * nq) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + m * N + n);
        for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(part + s * plane + m * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        float* o = out + m * ldo + n;
        *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(o) + v;
    }
}


// =================================================================================================
// Kernel "t256": 256x256 block tile, 8 waves (2 along M x 4 along N, 128x64 each = 4x2 MFMA tiles),
// K walked in 32-deep slabs through an NST-slot LDS ring (5 x 32 KiB = all 160 KiB of the CU by default).  Two wave groups (the two M
// halves) run the same program ONE BARRIER APART: in every barrier interval one group issues its
// 8 MFMAs (256 matrix-pipe cycles) at priority 1 while the other group issues its ds_read_b128
// fragment reads and 2 LDS-DMA pieces of the slab PD = NST-2 ahead.  Loads are never drained: one counted
// s_waitcnt vmcnt(4*(PD-1)) per slab retires the slab needed next while the newer ones stay in flight.
//
// Slab image: [256 A rows + 256 W rows][4 x 16-B chunks], chunk' = chunk ^ ((row>>2)&3) (applied on
// the LDS-DMA source address and on the fragment read) -> conflict-free ds_read_b128.
//
// Hazards (G = global barrier index; group 0 phase j: mid barrier 2j, end 2j+1; group 1: 2j+1, 2j+2):
//  RAW  slab s+1 is first read by group 0 after G = 4s+3; every wave waits vmcnt(4) for its slab-(s+1)
//       pieces before its phase-b mid barrier of slab s (G = 4s+2 / 4s+3).
//  WAR  slot (s+2)&3 last held slab s-2, whose last reads complete right after G = 4s-5; the first
//       DMA into it is issued after G = 4s-1.
// =================================================================================================
constexpr int T_BK = 32;
constexpr int T_SLAB = (T_BM + T_BN) * T_BK * 2;   // 32 KiB
constexpr int T_WOFF = T_BM * T_BK * 2;            // W rows start here inside a slab

// wait until at most `slabs` whole slabs (4 LDS-DMA pieces each, per wave) are still in flight
__device__ __forceinline__ void wait_slabs_in_flight(int slabs) {
    if (slabs <= 0) HX_WAIT_VM(0);
    else if (slabs == 1) HX_WAIT_VM(4);
    else if (slabs == 2) HX_WAIT_VM(8);
    else HX_WAIT_VM(12);
}

template <int EPI, int T_NST>
__global__ __launch_bounds__(512) void gemm_t256(GemmP p) {
    constexpr int PD = T_NST - 2;   // prefetch distance in slabs (slot being filled is never one being read)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int p_lo = xcd * p.ppx;
    int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
    if (np <= 0 || j >= np * p.nbn) return;
    const int grp = j / (GROUP_M * p.nbn);
    const int r = j - grp * GROUP_M * p.nbn;
    int gcount = np - grp * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
    const int nt_i = r / gcount, mt_i = p_lo + grp * GROUP_M + (r - nt_i * gcount);
    const int M0 = mt_i * T_BM, N0 = nt_i * T_BN;

    // LDS-DMA pieces: 1 KiB = 16 rows x 64 B; this wave owns A pieces 2w,2w+1 and W pieces 2w,2w+1
    const bf16_t* a_src[2];
    const bf16_t* w_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1;
        a_src[q] = p.A + (int64_t)gm * p.lda + chunk * 8;
        w_src[q] = p.W + (int64_t)gn * p.ldw + chunk * 8;
    }
    const int piece_off = wave * 2048;
    auto stage_a = [&](int s, int slot) {
        char* buf = smem + slot * T_SLAB + piece_off;
        glds16(a_src[0] + (int64_t)s * T_BK, buf);
        glds16(a_src[1] + (int64_t)s * T_BK, buf + 1024);
    };
    auto stage_w = [&](int s, int slot) {
        char* buf = smem + slot * T_SLAB + T_WOFF + piece_off;
        glds16(w_src[0] + (int64_t)s * T_BK, buf);
        glds16(w_src[1] + (int64_t)s * T_BK, buf + 1024);
    };

    // fragment read offsets (bytes inside a slab)
    const int frow = lane & 31, fsw = (frow >> 2) & 3, khalf = lane >> 5;
    const int koff0 = ((0 + khalf) ^ fsw) << 4, koff1 = ((2 + khalf) ^ fsw) << 4;
    const int a_base = (wr * 128 + frow) * (T_BK * 2);
    const int w_base = T_WOFF + (wc * 64 + frow) * (T_BK * 2);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    const int ns = p.K / T_BK;
    // ---- prologue: slabs 0..PD-1 in flight, slab 0 landed everywhere
    const int npro = ns < PD ? ns : PD;
    for (int s = 0; s < npro; ++s) { stage_a(s, s); stage_w(s, s); }
    wait_slabs_in_flight(npro - 1);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

    int slot_r = 0, slot_w = PD % T_NST;
    for (int s = 0; s < ns; ++s) {
        const char* buf = smem + slot_r * T_SLAB;
        const bool pre = s + PD < ns;
        // ================= phase a: rows [0,64) of this wave's 128 =================
        bf16x8 wf[2][2], af[2][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            wf[n][0] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 64 + koff0);
            wf[n][1] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 64 + koff1);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            af[m][0] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 64 + koff0);
            af[m][1] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 64 + koff1);
        }
        if (pre) stage_a(s + PD, slot_w);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        HX_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][ks], af[m][ks], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ================= phase b: rows [64,128) =================
        bf16x8 ag[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ag[m][0] = *reinterpret_cast<const bf16x8*>(buf + a_base + (2 + m) * 32 * 64 + koff0);
            ag[m][1] = *reinterpret_cast<const bf16x8*>(buf + a_base + (2 + m) * 32 * 64 + koff1);
        }
        if (pre) stage_w(s + PD, slot_w);
        {   // slab s+1 must have landed (this wave's pieces) before the next mid barrier
            int last = s + PD < ns - 1 ? s + PD : ns - 1;
            wait_slabs_in_flight(last - (s + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        HX_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[2 + m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][ks], ag[m][ks], acc[2 + m][n], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        slot_r = slot_r + 1 == T_NST ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == T_NST ? 0 : slot_w + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier

    // ---- epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = M0 + wr * 128 + i * 32 + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = N0 + wc * 64 + jn * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= p.N) continue;
                f32x4 v = {acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]};
                epilogue_store<EPI>(p, m, n, v);
            }
        }
    }
}


// =================================================================================================
// Kernel "t256p": same 256x256 tile / 8 waves / 32-deep slabs / XOR-swizzled slab image as t256, but
// software-pipelined INSIDE each wave instead of ping-ponging two wave groups: MFMA fragments are
// double-buffered in registers one phase ahead (ds_read_b128 of the next phase and the LDS-DMA of the
// slab PD = 3 ahead are issued under the current phase's 8 MFMAs), so the two waves of a SIMD simply
// share the matrix pipe and hide each other's memory instructions.  One barrier per slab:
//     vmcnt(4) lgkmcnt(0) BARRIER | 16 MFMA (slab s), each shadowing one of: 12 fragment reads of slab s+1,
//                                   4 LDS-DMA pieces of slab s+3
// Ring: 4 slots.  After barrier(s) every wave has finished all reads of slab s and slab s+1 has landed
// everywhere; DMA targets slot (s+3)&3 = slot of slab s-1, whose last reads finished before
// barrier(s-1).  vmcnt(4): the only pieces of this wave newer than slab s+1's are slab s+2's four.
// =================================================================================================
__device__ __forceinline__ void wait_vm_pieces(int n) {
    switch (n) {
        case 0: HX_WAIT_VM(0); break;
        case 2: HX_WAIT_VM(2); break;
        case 4: HX_WAIT_VM(4); break;
        case 6: HX_WAIT_VM(6); break;
        case 8: HX_WAIT_VM(8); break;
        default: HX_WAIT_VM(0); break;
    }
}


// -------------------------------------------------------------------------------------------------
// Epilogue of the 256x256 kernels: each wave transposes its 128x64 accumulator block through a
// private LDS staging area, 32 rows at a time, so that global traffic is whole 128-B lines
// (16 B per lane, 8 or 16 lanes per output row) instead of 8-B pieces scattered over 32 rows, and the
// bias / residual / pos operands are fetched in batches instead of one dependent load per store.
//   lane -> LDS : row (lane&31), 4 consecutive columns  (MFMA D layout with swapped operands)
//   LDS -> HBM  : bf16 out: 4 instr x (8 rows x 128 B);  f32 out: 8 instr x (4 rows x 256 B)
// -------------------------------------------------------------------------------------------------
constexpr int STG_RSB = 144;                 // bf16 staging row stride (bytes): 128 + 16 pad
constexpr int STG_RSF = 272;                 // f32 staging row stride: 256 + 16 pad
constexpr int STG_BYTES = 32 * STG_RSF;      // per wave

template <int EPI>
__device__ __forceinline__ void epilogue_256(const GemmP& p, f32x16 (&acc)[4][2], char* stg, int Mw, int Nw, int lane) {
    constexpr bool OUT_BF16 = (EPI == HIREST_EPI_BIAS_BF16 || EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_BIAS_QGELU_BF16);
    const int half = lane >> 5, lrow = lane & 31;
    f32x4 bv[2][4];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = Nw + jn * 32 + 8 * g + 4 * half;
            bv[jn][g] = (p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // ---- registers -> staging
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]};
                v += bv[jn][g];
                const int col = jn * 32 + 8 * g + 4 * half;
                if constexpr (OUT_BF16) {
                    if constexpr (EPI == HIREST_EPI_BIAS_GELU_BF16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    if constexpr (EPI == HIREST_EPI_BIAS_QGELU_BF16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                    *reinterpret_cast<bf16x4*>(stg + lrow * STG_RSB + col * 2) = o;
                } else {
                    *reinterpret_cast<f32x4*>(stg + lrow * STG_RSF + col * 4) = v;
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // ---- staging -> global, whole lines
        const int mb = Mw + i * 32;
        if constexpr (OUT_BF16) {
            bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), c = lane & 7;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * STG_RSB + c * 16);
                const int m = mb + r, n = Nw + c * 8;
                if (m < p.M) {
                    bf16_t* dst = outp + (int64_t)m * p.ldo + n;
                    if (n + 8 <= p.N) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
                    else if (n + 4 <= p.N) *reinterpret_cast<bf16x4*>(dst) = bf16x4{v[0], v[1], v[2], v[3]};
                }
            }
        } else {
            float* outp = reinterpret_cast<float*>(p.out);
            f32x4 v[8], o[8];
            int64_t off[8];
            bool ok[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + (lane >> 4), c = lane & 15;
                v[it] = *reinterpret_cast<const f32x4*>(stg + r * STG_RSF + c * 16);
                const int m = mb + r, n = Nw + c * 4;
                ok[it] = m < p.M && n < p.N;
                if constexpr (EPI == HIREST_EPI_PATCH_POS_F32) {
                    const int mm = ok[it] ? m : 0;
                    const int b = mm / p.P, pp = mm - b * p.P;
                    off[it] = ((int64_t)b * (p.P + 1) + 1 + pp) * p.ldo + n;
                    o[it] = ok[it] ? *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(1 + pp) * p.N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    off[it] = (int64_t)m * p.ldo + n;
                    if constexpr (EPI == HIREST_EPI_BIAS_RESID_F32)
                        o[it] = ok[it] ? *reinterpret_cast<const f32x4*>(outp + off[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                f32x4 w = v[it];
                if constexpr (EPI != HIREST_EPI_BIAS_F32) w += o[it];
                if (ok[it]) *reinterpret_cast<f32x4*>(outp + off[it]) = w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // staging rows are rewritten next pass
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_t256p(GemmP p) {
    constexpr int NST = 4, PD = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int p_lo = xcd * p.ppx;
    int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
    if (np <= 0 || j >= np * p.nbn) return;
    const int grp = j / (GROUP_M * p.nbn);
    const int r = j - grp * GROUP_M * p.nbn;
    int gcount = np - grp * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
    const int nt_i = r / gcount, mt_i = p_lo + grp * GROUP_M + (r - nt_i * gcount);
    const int M0 = mt_i * T_BM, N0 = nt_i * T_BN;

    const bf16_t* a_src[2];
    const bf16_t* w_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1;
        a_src[q] = p.A + (int64_t)gm * p.lda + chunk * 8;
        w_src[q] = p.W + (int64_t)gn * p.ldw + chunk * 8;
    }
    const int piece_off = wave * 2048;
    // DMA is issued UNCONDITIONALLY every phase (branch-free loop, constant vmcnt): past the last slab the
    // source is clamped to the last slab and the destination is an already-consumed slot.
    const int ns = p.K / T_BK;
    auto stage_a = [&](int s) {
        char* buf = smem + (s & (NST - 1)) * T_SLAB + piece_off;
        const int sc = s < ns ? s : ns - 1;
        glds16(a_src[0] + (int64_t)sc * T_BK, buf);
        glds16(a_src[1] + (int64_t)sc * T_BK, buf + 1024);
    };
    auto stage_w = [&](int s) {
        char* buf = smem + (s & (NST - 1)) * T_SLAB + T_WOFF + piece_off;
        const int sc = s < ns ? s : ns - 1;
        glds16(w_src[0] + (int64_t)sc * T_BK, buf);
        glds16(w_src[1] + (int64_t)sc * T_BK, buf + 1024);
    };

    const int frow = lane & 31, fsw = (frow >> 2) & 3, khalf = lane >> 5;
    const int koff0 = ((0 + khalf) ^ fsw) << 4, koff1 = ((2 + khalf) ^ fsw) << 4;
    const int a_base = (wr * 128 + frow) * (T_BK * 2);
    const int w_base = T_WOFF + (wc * 64 + frow) * (T_BK * 2);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    for (int s = 0; s < PD; ++s) { stage_a(s); stage_w(s); }
    HX_WAIT_VM(8);   // 4*(PD-1): slab 0 landed
    __builtin_amdgcn_s_barrier();

    // two named fragment sets (all 12 fragments of a slab: W 2x2, A 4x2), alternated by a 2x manual unroll
    struct Frags { bf16x8 w[2][2]; bf16x8 a[4][2]; };
    Frags f0, f1;
    auto load_frags = [&](const char* buf, Frags& f) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f.w[n][0] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 64 + koff0);
            f.w[n][1] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 64 + koff1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f.a[m][0] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 64 + koff0);
            f.a[m][1] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 64 + koff1);
        }
    };
    load_frags(smem, f0);

    auto slab = [&](int s, Frags& fc, Frags& fn) {
        // slab s+1 has landed for this wave (pieces newer than it: slab s+2's 4) -> publish with the barrier
        HX_WAIT_VM(4);
        HX_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // fragment reads of slab s+1, DMA of slab s+3 and this slab's 16 MFMAs; the sched_group_barrier
        // sequence below interleaves them (one memory instruction in each MFMA's shadow) instead of
        // letting all 8 waves burst 12 ds_read_b128 into the LDS queue right after the barrier.
        load_frags(smem + ((s + 1) & (NST - 1)) * T_SLAB, fn);   // (past the end: a stale slot, never used)
        stage_a(s + PD);
        stage_w(s + PD);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.w[n][ks], fc.a[m][ks], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (LDS-DMA piece)
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < ns; s += 2) {   // ns = K/32 is even (K % 64 == 0)
        slab(s, f0, f1);
        slab(s + 1, f1, f0);
    }
    HX_WAIT_VM(0);                  // no LDS-DMA may outlive the slab ring: it becomes the epilogue staging area
    HX_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();   // every wave is done reading slabs
    epilogue_256<EPI>(p, acc, smem + wave * STG_BYTES, M0 + wr * 128, N0 + wc * 64, lane);
}


// =================================================================================================
// Kernel "t256q": t256p with 64-deep K steps so that every LDS-DMA wave-instruction fetches whole
// 128-B lines (8 rows x 128 B).  Measured on MI355X (tools/probes/dma_probe.hip): LDS-DMA streaming of
// L2-resident rows tops out at 30 GB/s per CU with 64-B row segments and 47 GB/s per CU with 128-B
// segments — the 32-deep-slab kernel sat exactly on the former ceiling.
//
// Ring: 2 slots x 64 KiB (A 256 rows + W 256 rows, 128 B each, chunk' = chunk ^ ((row>>1)&7)).
// One iteration = one 32-deep half of a step (16 MFMAs per wave), fragments prefetched one iteration ahead:
//   it 2t   : [lgkmcnt(0) BARRIER]            16 MFMA (step t, k 0..31)  | read frags (step t, k 32..63)
//   it 2t+1 : [vmcnt(0) lgkmcnt(0) BARRIER]   16 MFMA (step t, k 32..63) | read frags (step t+1, k 0..31)
//                                                                        | DMA step t+2 -> slot t&1 (8 pieces)
// RAW: step t+1's pieces were issued during iteration 2t-1 and are waited for (vmcnt(0): nothing newer
// exists yet) before barrier(2t+1).  WAR: slot t&1 is last read during iteration 2t (k 32..63 of step t);
// those reads are complete before barrier(2t+1), the DMA is issued after it.
// =================================================================================================

template <int EPI>
__global__ __launch_bounds__(512) void gemm_t256q(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int p_lo = xcd * p.ppx;
    int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
    if (np <= 0 || j >= np * p.nbn) return;
    const int grp = j / (GROUP_M * p.nbn);
    const int r = j - grp * GROUP_M * p.nbn;
    int gcount = np - grp * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
    const int nt_i = r / gcount, mt_i = p_lo + grp * GROUP_M + (r - nt_i * gcount);
    const int M0 = mt_i * T_BM, N0 = nt_i * T_BN;

    // LDS-DMA pieces: 1 KiB = 8 rows x 128 B; this wave owns A pieces 4w..4w+3 and W pieces 4w..4w+3
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int gm = M0 + row; gm = gm < p.M ? gm : p.M - 1;
        int gn = N0 + row; gn = gn < p.N ? gn : p.N - 1;
        a_src[q] = p.A + (int64_t)gm * p.lda + chunk * 8;
        w_src[q] = p.W + (int64_t)gn * p.ldw + chunk * 8;
    }
    const int nst = p.K / Q_BK;
    const int piece_off = wave * 4096;
    auto stage = [&](int t) {   // DMA of step t (clamped past the end: dummy refetch into a consumed slot)
        char* buf = smem + (t & 1) * Q_STEP + piece_off;
        const int tc = t < nst ? t : nst - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(a_src[q] + (int64_t)tc * Q_BK, buf + q * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(w_src[q] + (int64_t)tc * Q_BK, buf + Q_WOFF + q * 1024);
    };

    const int frow = lane & 31, fsw = (frow >> 1) & 7, khalf = lane >> 5;
    const int a_base = (wr * 128 + frow) * (Q_BK * 2);
    const int w_base = Q_WOFF + (wc * 64 + frow) * (Q_BK * 2);
    int koff[4];   // chunk offsets for (h, ks): chunk = 4h + 2ks + khalf
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[c] = ((2 * c + khalf) ^ fsw) << 4;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    struct Frags { bf16x8 w[2][2]; bf16x8 a[4][2]; };
    Frags f0, f1;
    auto load_frags = [&](const char* buf, int h, Frags& f) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f.w[n][0] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 128 + koff[2 * h]);
            f.w[n][1] = *reinterpret_cast<const bf16x8*>(buf + w_base + n * 32 * 128 + koff[2 * h + 1]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f.a[m][0] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 128 + koff[2 * h]);
            f.a[m][1] = *reinterpret_cast<const bf16x8*>(buf + a_base + m * 32 * 128 + koff[2 * h + 1]);
        }
    };
    auto mfma16 = [&](Frags& fc) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc.w[n][ks], fc.a[m][ks], acc[m][n], 0, 0, 0);
    };

    // ---- prologue: steps 0 and 1 in flight, step 0 landed, first fragments in registers
    stage(0);
    stage(1);
    HX_WAIT_VM(8);
    __builtin_amdgcn_s_barrier();
    load_frags(smem, 0, f0);

    // Waves whose 64 output columns (or 128 rows) lie entirely in the padding of a ragged edge tile
    // (N = 1408 = 5.5 tiles: half the waves of every 6th tile) skip fragments and MFMAs — they only keep
    // the DMA and barrier schedule.  The chip is power-limited here, so idle matrix pipes are not wasted.
    const bool active = (N0 + wc * 64 < p.N) && (M0 + wr * 128 < p.M);
    if (active) {
        for (int t = 0; t < nst; ++t) {
            const char* cur = smem + (t & 1) * Q_STEP;
            const char* nxt = smem + ((t + 1) & 1) * Q_STEP;
            // ---- iteration 2t: k 0..31 of step t
            HX_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            load_frags(cur, 1, f1);
            mfma16(f0);
    #pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- iteration 2t+1: k 32..63 of step t
            HX_WAIT_VM(0);
            HX_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            load_frags(nxt, 0, f0);          // (past the end: stale data, never used)
            stage(t + 2);
            mfma16(f1);
            // (the DMA writes LDS, so the compiler keeps it behind the fragment reads: reads first, then DMA)
    #pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
    #pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int t = 0; t < nst; ++t) {
            HX_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
            HX_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
            stage(t + 2);
        }
    }
    HX_WAIT_VM(0);
    HX_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    epilogue_256<EPI>(p, acc, smem + wave * STG_BYTES, M0 + wr * 128, N0 + wc * 64, lane);
}


// =================================================================================================
// Kernel "p256": PERSISTENT version of t256q.  One workgroup per CU walks its XCD's tile list; the LDS-DMA
// stream is continuous across tiles (while the last two K steps of tile i are multiplied, the first two steps
// of tile i+1 are already landing in the ring), so there is no per-tile prologue, no dummy refetch at the end
// of a tile and no workgroup launch / drain per tile.  Other changes against t256q:
//   * one barrier per 64-deep step (the barrier in front of the first half-step protected nothing);
//   * the epilogue stages through a wave-private 4-KiB XOR-swizzled area BEHIND the ring (ring 128 KiB +
//     staging = 160 KiB for 8 waves), so the ring keeps receiving the next tile during the epilogue;
//   * DMA addresses are (uniform tile base + k) in SGPRs + one 32-bit per-lane offset per piece;
//   * 8-wave form: the two waves of a SIMD (w, w+4) own different column halves, so in a ragged N edge tile
//     (N = 1408 = 5.5 tiles) every SIMD keeps one active wave and the tile takes half the time;
//   * WN = 128 gives the 4-wave form (2x2 waves of 128x128, accumulators in AGPRs): 2/3 of the LDS fragment
//     reads per FLOP.
// Comparison that motivated it (rocprofv3 PMC, fc1 shape, same clocks ~1.6 GHz): hipBLASLt's 256x256x64
// stream-K kernel keeps the matrix pipe 72 % busy, t256q 58 %.
// =================================================================================================
template <int EPI, int WN, bool DBG, int RD = 1>   // RD: residual look-ahead of the LN-statistics epilogue (gemm_shared.h)
__global__ __launch_bounds__(WN == 64 ? 512 : 256) void gemm_p256(GemmP p) {
    // timing-experiment switches (hirest_gemm_debug_mode) exist only in the DBG instantiation: a branch inside the K loop
    // splits the scheduling region and destroys the MFMA / ds_read / LDS-DMA interleave
    const int dbg = DBG ? p.dbg : 0, stagger = DBG ? p.stagger : 0;
    constexpr int NW = 512 / WN;        // 8 or 4 waves: 2 (M) x NW/2 (N)
    constexpr int NI = WN / 16;         // 16-column MFMA tiles per wave (4 or 8); 8 16-row tiles
    constexpr int PPW = 32 / NW;        // LDS-DMA pieces per operand per wave per step
    constexpr int NR = 8 + NI;          // fragment reads per 32-deep half-step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wr, wc;
    if constexpr (NW == 8) { wr = (wave >> 1) & 1; wc = (wave & 1) + 2 * (wave >> 2); }
    else { wr = wave >> 1; wc = wave & 1; }

    // ---- this block's tile list: XCD (bid & 7) owns M panels [p_lo, p_lo+np); its CUs take every nslot-th tile
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nslot = gridDim.x >> 3;
    int p_lo, np;
    xcd_panels<DBG>(p, xcd, p_lo, np);
    if (np <= 0) return;
    // Work list of the XCD = units, unit j goes to CU slot j % nslot.  The ~32 tiles in flight form a patch (a panels x
    // b column tiles) whose operand lines are shared through the XCD's L2:
    //   * few column tiles (N = 1408: 6): panel-major, all column tiles of a panel together;
    //   * many column tiles: groups of GROUP_M panels, column-major inside a group (8 panels x 4 columns per round).
    // A unit is one 256x256 tile.  Experiment kept behind debug bit 5: when N % 256 <= 128 the ragged last column tile
    // keeps one wave per SIMD busy and takes half the time; pairing two of them into one unit makes all units equal so
    // the CUs move in lock-step.  FETCH_SIZE did not drop (fc2: 25 GB per launch either way — what the schedule must
    // pull through a 4-MB L2: 5.3 A panels + 6 W tiles of 3 MB per round) and time did not improve, so it is off.
    const bool half_edge = (dbg & 32) && (p.N % T_BN) != 0 && (p.N % T_BN) <= T_BN / 2;
    const int ncf = half_edge ? p.nbn - 1 : p.nbn;                       // column tiles that take full time
    const bool panel_major = (dbg & 8) ? false : (dbg & 16) ? true : p.nbn <= 8;   // dbg bits: A/B timing of the order
    const int upp = 2 * ncf + (half_edge ? 1 : 0);                       // units per panel pair (panel-major)
    const int ugf = GROUP_M * ncf + (half_edge ? GROUP_M / 2 : 0);       // units per full panel group (grouped)
    int nunit;
    if (panel_major) nunit = (np >> 1) * upp + ((np & 1) ? ncf + (half_edge ? 1 : 0) : 0);
    else { const int rem = np % GROUP_M; nunit = (np / GROUP_M) * ugf + (rem ? rem * ncf + (half_edge ? (rem + 1) / 2 : 0) : 0); }
    if (slot >= nunit) return;
    if (stagger) {   // timing experiment: de-synchronise the CUs so that their HBM-heavy epilogues do not coincide
        const int who = stagger == 1 ? (slot & 3) : stagger == 2 ? (xcd & 3) : ((slot + xcd) & 7);
        const int units = who * (stagger == 3 ? (p.K / Q_BK + 15) / 16 : (p.K / Q_BK + 7) / 8);
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // tile `sub` of unit j -> origin; returns the number of tiles in the unit (1 or 2)
    auto unit_tile = [&](int j, int sub, int& M0, int& N0) -> int {
        if (p.rev) j = nunit - 1 - j;                                    // (not combined with the paired-edge experiment)
        int panel, col, cnt = 1;
        if (panel_major) {
            const int full_pairs = np >> 1;
            if (j >= full_pairs * upp) {                                 // the odd last panel
                const int r = j - full_pairs * upp;
                panel = np - 1; col = r;                                 // r == ncf is its (single) edge tile
            } else {
                const int pair = j / upp, r = j - pair * upp;
                if (r < ncf) { panel = 2 * pair; col = r; }
                else if (r < 2 * ncf) { panel = 2 * pair + 1; col = r - ncf; }
                else { panel = 2 * pair + sub; col = ncf; cnt = 2; }
            }
        } else {
            int grp = j / ugf, r = j - grp * ugf, gcount = GROUP_M;
            if (grp >= np / GROUP_M) { grp = np / GROUP_M; r = j - grp * ugf; gcount = np - grp * GROUP_M; }
            if (r < gcount * ncf) {
                col = r / gcount; panel = grp * GROUP_M + (r - col * gcount);
            } else {
                const int e = r - gcount * ncf;
                col = ncf; panel = grp * GROUP_M + 2 * e + sub;
                cnt = (2 * e + 1 < gcount) ? 2 : 1;
            }
        }
        M0 = (p_lo + panel) * T_BM; N0 = col * T_BN;
        return cnt;
    };

    // ---- LDS-DMA stream state (runs up to two K steps ahead of the MFMAs, across tile boundaries)
    const int nst = p.K / Q_BK;
    uint32_t a_off[PPW], w_off[PPW];
    const char* a_base; const char* w_base;
    auto set_sources = [&](int M0, int N0) {
        if (dbg & 4) { M0 = 0; N0 = 0; }      // timing experiment: every tile streams tile (0,0)'s operands (L2-resident)
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int row = (wave * PPW + q) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            int ra = p.M - 1 - M0; ra = row < ra ? row : ra;
            int rn = p.N - 1 - N0; rn = row < rn ? row : rn;
            a_off[q] = (uint32_t)(ra * (int)p.lda + chunk * 8) * 2u;
            w_off[q] = (uint32_t)(rn * (int)p.ldw + chunk * 8) * 2u;
        }
        a_base = reinterpret_cast<const char*>(p.A + (int64_t)M0 * p.lda);
        w_base = reinterpret_cast<const char*>(p.W + (int64_t)N0 * p.ldw);
    };
    int dma_j = slot, dma_sub = 0, dma_cnt = 1, dma_k = 0, dma_g = 0;
    bool dma_live = true;
    auto stage = [&]() {   // issue the next step of the stream into ring slot dma_g & 1
        if (dbg & 1) return;                  // timing experiment: no LDS-DMA in the loop
        char* buf = smem + (dma_g & 1) * Q_STEP + wave * (PPW * 1024);
        const char* ab = a_base + (int64_t)dma_k * (Q_BK * 2);
        const char* wb = w_base + (int64_t)dma_k * (Q_BK * 2);
#pragma unroll
        for (int q = 0; q < PPW; ++q) glds16(ab + a_off[q], buf + q * 1024);
        if (dbg & 128) return;                // timing experiment: A operand only (half the LDS-DMA traffic)
#pragma unroll
        for (int q = 0; q < PPW; ++q) glds16(wb + w_off[q], buf + Q_WOFF + q * 1024);
    };
    auto advance = [&]() {   // wave-uniform; past the last tile the stream refetches its last step (never consumed)
        ++dma_g;
        if (dma_live && ++dma_k == nst) {
            if (dma_sub + 1 < dma_cnt) ++dma_sub; else { dma_j += nslot; dma_sub = 0; }
            if (dma_j < nunit) { int m0, n0; dma_cnt = unit_tile(dma_j, dma_sub, m0, n0); set_sources(m0, n0); dma_k = 0; }
            else { dma_live = false; dma_k = nst - 1; }
        }
    };

    // ---- fragments of v_mfma_f32_16x16x32_bf16: lane -> row (lane & 15), 8 consecutive k at 8 * (lane >> 4)
    const int frow = lane & 15, fsw = (frow >> 1) & 7, kg = lane >> 4;
    const int a_frag = (wr * 128 + frow) * (Q_BK * 2);
    const int w_frag = Q_WOFF + (wc * WN + frow) * (Q_BK * 2);
    int koff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) koff[h] = ((4 * h + kg) ^ fsw) << 4;

    struct Frags { bf16x8 w[NI]; bf16x8 a[8]; };
    Frags f0, f1;
    auto load_frags = [&](const char* buf, int h, Frags& f) {
#pragma unroll
        for (int n = 0; n < NI; ++n) f.w[n] = *reinterpret_cast<const bf16x8*>(buf + w_frag + n * 16 * 128 + koff[h]);
#pragma unroll
        for (int m = 0; m < 8; ++m) f.a[m] = *reinterpret_cast<const bf16x8*>(buf + a_frag + m * 16 * 128 + koff[h]);
    };
    f32x4 acc[8][NI];
    auto mfma_half = [&](Frags& fc) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc.w[n], fc.a[m], acc[m][n], 0, 0, 0);
    };

    // ---- prologue of the stream: steps 0 and 1
    {
        int m0, n0;
        dma_cnt = unit_tile(slot, 0, m0, n0);
        set_sources(m0, n0);
    }
    stage(); advance();
    HX_WAIT_VM(0);                       // step 0 landed
    stage(); advance();
    __builtin_amdgcn_s_barrier();

    char* stg = smem + 2 * Q_STEP + wave * p_stg_bytes(EPI);
    int g = 0;   // global step index of the MFMA side: step g lives in ring slot g & 1
    for (int j = slot, sub = 0; j < nunit;) {
        int M0, N0;
        const int cnt = unit_tile(j, sub, M0, N0);
        if (sub + 1 < cnt) ++sub; else { j += nslot; sub = 0; }           // cursor now points at the next tile
        const bool active = (N0 + wc * WN < p.N) && (M0 + wr * 128 < p.M);
        if (active) {
            if constexpr (epi_is_lnfold(EPI)) {   // (mean, rstd) of this wave's 128 rows: global -> LDS by DMA, lands under the K loop
                int r0 = M0 + wr * 128 + 2 * lane;
                const int last = (p.M - 1) & ~1;   // the stats buffer is padded to an even number of rows
                r0 = r0 < last ? r0 : last;
                glds16(reinterpret_cast<const float*>(p.aux0) + 2 * (int64_t)r0, stg + P_STG);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            load_frags(smem + (g & 1) * Q_STEP, 0, f0);
            for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                const char* nxt = smem + ((g + 1) & 1) * Q_STEP;
                // ---- first half: k 0..31 of step g | read the second half's fragments
                __builtin_amdgcn_sched_barrier(0);
                if (!(dbg & 256)) load_frags(cur, 1, f1);   // (bit8, timing experiment: half the fragment reads)
                mfma_half(f0);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- second half: step g+1 has landed everywhere and nobody reads slot g&1 any more after the
                // barrier -> read step g+1's first fragments, refill slot g&1 with step g+2, k 32..63 of step g
                if (!(dbg & 64)) HX_WAIT_VM(0);   // (bit6, timing experiment: do not wait for the LDS-DMA)
                HX_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // The refill goes out FIRST: its round trip (L2 / Infinity Cache / HBM) is the longest latency of the step
                // and it has exactly one step to land (fc2, whose A panels stream from HBM: 4.5 -> 4.2 ms).  It targets
                // slot g&1; the fragment reads below are from the other slot.
                stage();
                load_frags(nxt, 0, f0);          // (at the end of a tile: the next tile's step 0 — reloaded below)
                mfma_half(f1);
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * PPW, 0);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                advance();
            }
            epilogue_p<EPI, NI, epi_is_lnfold(EPI), 8, RD, false, DBG>(p, acc, stg, M0 + wr * 128, N0 + wc * WN, lane);
        } else {
            for (int t = 0; t < nst; ++t, ++g) {
                HX_WAIT_VM(0);
                __builtin_amdgcn_s_barrier();
                stage();
                advance();
            }
        }
    }
    HX_WAIT_VM(0);
}


// =================================================================================================
// Kernel "pp256": persistent 256x256 tile, PING-PONG wave groups.  Waves 0-3 (group 0) and 4-7 (group 1, the same
// SIMDs) run the same phase sequence one barrier apart, so that while one wave of a SIMD executes its 16 MFMAs the
// other one reads fragments, issues LDS-DMA and waits — the matrix pipe always has exactly one feeder.
// A 64-deep K step is four phases over the wave's 128x64 output, one 64x32 quadrant each over the FULL K step:
//     P1 (a0,b0)   P2 (a0,b1)   P3 (a1,b1)   P4 (a1,b0)      a = 64-row half of the wave's A block, b = 32-column half of its W block
//     phase = [ LOAD: vmcnt | ds_read | LDS-DMA ]  barrier  lgkmcnt(0)  [ 16 MFMA ]  barrier
// Each 128-row group of a ring slot is read in ONE LOAD (a0,b0: P1; b1: P2; a1: P3) and refilled two phases later
// (a0,b0: P3; b1: P4; a1: next P1) with the data of step +2; a group is waited for (counted vmcnt: "all but the 8
// newest pieces") in the LOAD one phase BEFORE the LOAD that reads it (a0,b0: P4; b1: P1; a1: P2).
// Why that is race-free with the one-barrier skew (hardware barrier k pairs group 0's k-th with group 1's (k-1)-th):
//   RAW  a wave's wait in LOAD(p-1) precedes its next barrier; every reader passes one more barrier before its
//        LOAD(p), and by then the other group has arrived at (= executed everything before) the paired barrier, which
//        lies after its own LOAD(p-1).
//   WAR  reads issued in LOAD(p) retire at the lgkmcnt(0) right after the phase's first barrier; a refill in LOAD(p+2)
//        of either group comes after a barrier at which the other group had already passed that lgkmcnt(0).
// Registers: one A sub-block (8 fragments) + both W sub-blocks (2 x 4): 64 VGPRs — nothing is prefetched across phases.
// =================================================================================================
// X3 ("bf16x3", reference-rank precision at 3/16 of the fp32-MFMA cost): both operands are fp32 values split into bf16 hi + lo
// parts, stored interleaved along K in blocks of 64 = [hi of 32 consecutive k | lo of the same 32 k] (hirest_split2_bf16), so one
// 64-deep step of the ring holds hi and lo of BOTH operands for 32 real k, and a phase issues three MFMA groups on the fragments
// it has already read — W_hi A_lo, W_lo A_hi, W_hi A_hi (small terms first; W_lo A_lo, 2^-16 of the product, is dropped) —
// instead of two: 1.5 x the matrix work per LDS byte of the plain kernel, nothing else changes (same ring, barriers, epilogues).
// PH2 (round 4): TWO phases per 64-deep step instead of four — Q1 = (a0; b0, b1), Q2 = (a1; b1, b0), 32 MFMAs each (48 with X3) — on the same
// ring, fragments and DMA pieces.  Why: the X3 form showed that a phase pair costs ~544 cycles + its MFMA issue time whatever the MFMA count
// (DESIGN 4.1f), i.e. a per-phase cost (two workgroup barriers, the LDS round trip of the phase's fragments, the MFMA pipe's fill) that
// longer phases amortise.  Differences from the four-phase schedule: a phase's fragment reads are RETIRED (lgkmcnt(0)) before its first
// barrier — they run under the other group's 32 MFMAs — so a region may be refilled in the very next LOAD of either group:
//     Q1 LOAD(g): wait own a1(g) pieces [vmcnt(6)] | read a0, b0, b1 of slot g & 1 | issue a1(g + 1) (pending group) | lgkmcnt(0)
//     Q2 LOAD(g): wait own a0 / b0 / b1(g + 1) pieces [vmcnt(2)] | read a1 | issue a0, b0, b1 of step g + 2 into slot g & 1 | lgkmcnt(0)
//   RAW  a wave's counted wait for the pieces of a group lies one LOAD before the LOAD that reads the group; every reader passes at least
//        one barrier in between at which the other group had executed that wait (hardware barrier k pairs group 0's k-th with group 1's
//        (k - 1)-th: group 0's Q1 LOAD(g) follows hardware barrier 4 g - 1, group 1's Q2 LOAD(g - 1) — its wait — precedes it).
//   WAR  reads of LOAD(p) are retired before that phase's first barrier; the refill is issued in LOAD(p + 1), which for group 0 follows
//        hardware barrier 2 p + 1 (group 1's LOAD(p) precedes it) and for group 1 follows 2 p + 2.
// DBG: the walk / timing switches of hirest_gemm_debug_mode (team walk, A wrap, staggered start, epilogue knock-outs, old XCD split) exist
// only in the gemm_pq256_dbg instantiations; every production kernel is compiled with them folded away.
template <int EPI, int RD, bool X3, int PH2 = 0, bool DBG = false>      // PH2: 0 four phases per step, 1 two phases (issuing the LDS-DMA before the fragment reads
                                                      // of a LOAD instead of after them measured 0.3-0.7 % slower: not kept)
__device__ __forceinline__ void pp256_body(const GemmP& p) {
    constexpr int NI = 4;
    const int sched = DBG ? p.sched : 0, stagger = DBG ? p.stagger : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                       // ping-pong group; waves w and w+4 share a SIMD
    const int wr = grp, wc = wave & 3;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nslot = gridDim.x >> 3;
    int p_lo, np;
    xcd_panels<DBG>(p, xcd, p_lo, np);
    if (np <= 0) return;
    const bool panel_major = p.nbn <= 8;
    const int nunit = np * p.nbn;
    if (slot >= nunit) return;
    // Units of this CU slot.  Default: unit j -> slot j % nslot, i.e. slot, slot + nslot, ...  With few column tiles (fc2 / proj: nbn = 6 of
    // 32 slots) that makes the set of CUs sharing an A panel change every round (32 / 6 is not an integer) and lets them drift apart in
    // K, so the panel's slabs are pulled through the XCD's 4-MB L2 again and again (fc2: 17.6 GB fetched per launch for 4.7 algorithmic).
    // TEAM walk (p.sched bit 2 = hirest_gemm_debug_mode bit 18, A/B): the slots form nslot / nbn fixed teams of nbn CUs — team t takes
    // panels t, t + teams, ... one column tile per member, so the same nbn CUs start every panel together and stay K-aligned — and the
    // nslot % nbn slots left over each walk whole panels of the tail of the XCD's range on their own, column after column.
    const bool team_walk = DBG && (sched & 4) && panel_major && p.nbn <= nslot && nunit >= nslot;
    const int teams = team_walk ? nslot / p.nbn : 0, tslots = teams * p.nbn, solo = nslot - tslots;
    const int np_solo = team_walk && solo ? (np * solo + nslot / 2) / nslot : 0, np_team = np - np_solo;
    int count;
    if (!team_walk) count = (nunit - slot + nslot - 1) / nslot;
    else if (slot < tslots) { const int tm = slot / p.nbn; count = tm < np_team ? (np_team - tm + teams - 1) / teams : 0; }
    else { const int sl_ = slot - tslots; count = sl_ < np_solo ? ((np_solo - sl_ + solo - 1) / solo) * p.nbn : 0; }
    if (count <= 0) return;
    if (stagger) {   // timing experiment: de-synchronise the CUs so that their HBM-heavy epilogues do not coincide
        const int who = stagger == 1 ? (slot & 3) : stagger == 2 ? (xcd & 3) : ((slot + xcd) & 7);
        const int units = who * (stagger == 3 ? (p.K / Q_BK + 15) / 16 : (p.K / Q_BK + 7) / 8);
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }
    auto team_origin = [&](int i, int& M0, int& N0) {        // i-th unit of this slot under the team walk
        if (p.rev) i = count - 1 - i;
        int panel, col;
        if (slot < tslots) { panel = slot / p.nbn + i * teams; col = slot % p.nbn; }
        else { panel = np_team + (slot - tslots) + (i / p.nbn) * solo; col = i % p.nbn; }
        M0 = (p_lo + panel) * T_BM; N0 = col * T_BN;
    };
    auto tile_origin = [&](int j, int& M0, int& N0) {        // same walk as p256 (see there); j = slot + i nslot
        if (team_walk) { team_origin((j - slot) / nslot, M0, N0); return; }
        if (p.rev) j = nunit - 1 - j;
        if (panel_major) {
            const int mt_i = j / p.nbn;
            M0 = (p_lo + mt_i) * T_BM; N0 = (j - mt_i * p.nbn) * T_BN;
            return;
        }
        const int g_ = j / (GROUP_M * p.nbn);
        const int r = j - g_ * GROUP_M * p.nbn;
        int gcount = np - g_ * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
        const int nt_i = r / gcount, mt_i = p_lo + g_ * GROUP_M + (r - nt_i * gcount);
        M0 = mt_i * T_BM; N0 = nt_i * T_BN;
    };

    // ---- LDS-DMA stream.  Row group -> 16 pieces of 8 rows, two per wave: piece j = 2*wave + q.
    //   A groups: j < 8 -> rows a*64 + 8j of the wr = 0 block, j >= 8 -> of the wr = 1 block (+128)
    //   W groups: run j/4 = wc block, rows wc*64 + b*32 + 8*(j%4)
    const int nst = p.K / Q_BK;
    uint32_t a_off[2][2], w_off[2][2];
    int a_lds[2][2], w_lds[2][2];
    const char* a_base; const char* w_base;
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * wave + q;
            a_lds[hsel][q] = ((j >> 3) * 128 + hsel * 64 + (j & 7) * 8) * 128;
            w_lds[hsel][q] = Q_WOFF + ((j >> 2) * 64 + hsel * 32 + (j & 3) * 8) * 128;
        }
    auto set_sources = [&](int M0, int N0) {
#pragma unroll
        for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ra0 = (a_lds[hsel][q] >> 7) + (lane >> 3);
                const int rw0 = ((w_lds[hsel][q] - Q_WOFF) >> 7) + (lane >> 3);
                const int ca = (lane & 7) ^ ((ra0 >> 1) & 7), cw = (lane & 7) ^ ((rw0 >> 1) & 7);
                int ra = p.M - 1 - M0; ra = ra0 < ra ? ra0 : ra;
                int rn = p.N - 1 - N0; rn = rw0 < rn ? rw0 : rn;
                a_off[hsel][q] = (uint32_t)(ra * (int)p.lda + ca * 8) * 2u;
                w_off[hsel][q] = (uint32_t)(rn * (int)p.ldw + cw * 8) * 2u;
            }
        // TIMING EXPERIMENT (p.sched bit 3 = hirest_gemm_debug_mode bit 19; results are wrong): the A operand of every tile wraps into the
        // XCD's first four panels (2.9 MB: L2-resident after the first touch) — what a perfect L2 hit rate on A would be worth
        const int Ma = (DBG && (sched & 8)) ? (p_lo + ((M0 / T_BM - p_lo) & 3)) * T_BM : M0;
        a_base = reinterpret_cast<const char*>(p.A + (int64_t)Ma * p.lda);
        w_base = reinterpret_cast<const char*>(p.W + (int64_t)N0 * p.ldw);
    };
    // The refill stream is two steps ahead of the step being multiplied; its a1 group is issued one step later than its
    // a0 / b0 / b1 groups (in the next step's P1), so the stream keeps the previous step's A source as well.
    int dma_j = slot, dma_k = 0, dma_g = 0;
    bool dma_live = true;
    const char* a1_base; uint32_t a1_off[2]; int a1_k = 0, a1_g = 0;    // pending a1 group: source of the step issued last
    auto issue_a0 = [&]() {
        char* buf = smem + (dma_g & 1) * Q_STEP;
        const char* ab = a_base + (int64_t)dma_k * (Q_BK * 2);
        glds16(ab + a_off[0][0], buf + a_lds[0][0]);
        glds16(ab + a_off[0][1], buf + a_lds[0][1]);
    };
    auto issue_w = [&](int hsel) {
        char* buf = smem + (dma_g & 1) * Q_STEP;
        const char* wb = w_base + (int64_t)dma_k * (Q_BK * 2);
        glds16(wb + w_off[hsel][0], buf + w_lds[hsel][0]);
        glds16(wb + w_off[hsel][1], buf + w_lds[hsel][1]);
    };
    auto issue_a1_pending = [&]() {
        char* buf = smem + (a1_g & 1) * Q_STEP;
        const char* ab = a1_base + (int64_t)a1_k * (Q_BK * 2);
        glds16(ab + a1_off[0], buf + a_lds[1][0]);
        glds16(ab + a1_off[1], buf + a_lds[1][1]);
    };
    auto advance = [&]() {   // after a step's a0, b0, b1 were issued: remember its a1 group, move to the next step
        a1_base = a_base; a1_off[0] = a_off[1][0]; a1_off[1] = a_off[1][1]; a1_k = dma_k; a1_g = dma_g;
        ++dma_g;
        if (dma_live && ++dma_k == nst) {
            dma_j += nslot;
            if (dma_j < slot + count * nslot) { int m0, n0; tile_origin(dma_j, m0, n0); set_sources(m0, n0); dma_k = 0; }
            else { dma_live = false; dma_k = nst - 1; }        // past the end: refetch of the last step, never consumed
        }
    };

    // ---- fragments of v_mfma_f32_16x16x32_bf16: lane -> row (lane & 15), 8 consecutive k at 8 * (lane >> 4)
    const int frow = lane & 15, fsw = (frow >> 1) & 7, kg = lane >> 4;
    const int a_frag = (wr * 128 + frow) * (Q_BK * 2);
    const int w_frag = Q_WOFF + (wc * 64 + frow) * (Q_BK * 2);
    int koff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) koff[h] = ((4 * h + kg) ^ fsw) << 4;
    struct FA { bf16x8 v[4][2]; };
    struct FW { bf16x8 v[2][2]; };
    FA Af;
    FW W0, W1;
    auto read_a = [&](const char* buf, int a) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) Af.v[m][h] = *reinterpret_cast<const bf16x8*>(buf + a_frag + (a * 64 + m * 16) * 128 + koff[h]);
    };
    auto read_w = [&](const char* buf, int b, FW& f) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) f.v[n][h] = *reinterpret_cast<const bf16x8*>(buf + w_frag + (b * 32 + n * 16) * 128 + koff[h]);
    };
    f32x4 acc[8][NI];
    auto mfma_q = [&](int a, int b, FW& fw) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (X3) {
#pragma unroll
            for (int g3 = 0; g3 < 3; ++g3)                       // (W half, A half): (hi, lo), (lo, hi), (hi, hi)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[a * 4 + m][b * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw.v[n][g3 == 1 ? 1 : 0], Af.v[m][g3 == 0 ? 1 : 0],
                                                                                             acc[a * 4 + m][b * 2 + n], 0, 0, 0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[a * 4 + m][b * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw.v[n][h], Af.v[m][h], acc[a * 4 + m][b * 2 + n], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };

    // ---- prologue: steps 0 and 1 completely (their a1 groups included), everything drained; then the skew barrier
    {
        int m0, n0;
        tile_origin(slot, m0, n0);
        set_sources(m0, n0);
    }
    for (int i = 0; i < 2; ++i) { issue_a0(); issue_w(0); issue_w(1); advance(); issue_a1_pending(); }
    // the stream's "pending a1" now belongs to step 1 and is already issued: the first in-loop P1 must not issue it again.
    // Make the pending group the a1 group of step 2 instead by issuing step 2's early groups in the loop as usual:
    // (P1 issues the pending a1 = step g+1's; see the loop) -> mark it consumed
    bool a1_valid = false;
    HX_WAIT_VM(0);
    bar();
    if (grp == 1) bar();                              // group 1 runs one barrier behind group 0 from here on

    char* stg = smem + 2 * Q_STEP + wave * p_stg_bytes(EPI);
    int g = 0;                                        // global step index of the MFMA side: step g lives in ring slot g & 1
    for (int j = slot; j < slot + count * nslot; j += nslot) {
        int M0, N0;
        tile_origin(j, M0, N0);
        const bool active = (N0 + wc * 64 < p.N) && (M0 + wr * 128 < p.M);
        if (active) {
            if constexpr (epi_is_lnfold(EPI)) {
                // (mean, rstd) of this wave's 128 rows and the bias | column-sum slices of its 64 columns: global -> LDS by DMA, two pieces that
                // are OLDER than everything the K loop issues, so its counted waits cover them (the second LOAD of step 0 retires them) and the
                // epilogue starts without a global load (round 5: two exposed round trips per tile, vmcnt(0) each, in front of fc1 / qkv's epilogue)
                int r0 = M0 + wr * 128 + 2 * lane;
                const int last = (p.M - 1) & ~1;   // the stats buffer is padded to an even number of rows
                r0 = r0 < last ? r0 : last;
                glds16(reinterpret_cast<const float*>(p.aux0) + 2 * (int64_t)r0, stg + P_STG);
                int c = N0 + wc * 64 + 4 * (lane & 15);
                c = c < p.N - 4 ? c : p.N - 4;
                const float* src = ((lane & 16) || !p.bias) ? reinterpret_cast<const float*>(p.aux1) : p.bias;
                glds16(src + c, stg + P_BCS);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (active && PH2) {
            for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                // ---- Q1 (a0; b0, b1)
                HX_WAIT_VM(6);
                __builtin_amdgcn_sched_barrier(0);
                read_a(cur, 0); read_w(cur, 0, W0); read_w(cur, 1, W1);
                if (a1_valid) issue_a1_pending();
                HX_WAIT_LGKM0();
                bar();
                mfma_q(0, 0, W0);
                mfma_q(0, 1, W1);
                bar();
                // ---- Q2 (a1; b1, b0)
                HX_WAIT_VM(2);
                __builtin_amdgcn_sched_barrier(0);
                read_a(cur, 1);
                issue_a0(); issue_w(0); issue_w(1);
                advance();
                a1_valid = true;
                HX_WAIT_LGKM0();
                bar();
                mfma_q(1, 1, W1);
                mfma_q(1, 0, W0);
                bar();
            }
            epilogue_p<EPI, NI, epi_is_lnfold(EPI), 8, RD, false, DBG, epi_is_lnfold(EPI)>(p, acc, stg, M0 + wr * 128, N0 + wc * 64, lane);
        } else if (PH2) {
            for (int t = 0; t < nst; ++t, ++g) {             // the same wait / DMA / barrier skeleton for a wave in the padding of an edge tile
                HX_WAIT_VM(6);
                if (a1_valid) issue_a1_pending();
                bar(); bar();
                HX_WAIT_VM(2);
                issue_a0(); issue_w(0); issue_w(1);
                advance();
                a1_valid = true;
                bar(); bar();
            }
        } else if (active) {
            for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                // ---- P1 (a0,b0): wait b1(g); read a0, b0; refill a1 of step g+1 (its rows were last read in P3 of step g-1)
                HX_WAIT_VM(8);
                __builtin_amdgcn_sched_barrier(0);
                read_a(cur, 0); read_w(cur, 0, W0);
                if (a1_valid) issue_a1_pending();
                bar();
                HX_WAIT_LGKM0();
                mfma_q(0, 0, W0);
                bar();
                // ---- P2 (a0,b1): wait a1(g); read b1
                HX_WAIT_VM(8);
                __builtin_amdgcn_sched_barrier(0);
                read_w(cur, 1, W1);
                bar();
                HX_WAIT_LGKM0();
                mfma_q(0, 1, W1);
                bar();
                // ---- P3 (a1,b1): read a1; refill a0, b0 with step g+2 (read in P1)
                __builtin_amdgcn_sched_barrier(0);
                read_a(cur, 1);
                issue_a0(); issue_w(0);
                bar();
                HX_WAIT_LGKM0();
                mfma_q(1, 1, W1);
                bar();
                // ---- P4 (a1,b0): wait a0, b0 of step g+1; refill b1 with step g+2 (read in P2)
                HX_WAIT_VM(8);
                __builtin_amdgcn_sched_barrier(0);
                issue_w(1);
                advance();
                a1_valid = true;
                bar();
                mfma_q(1, 0, W0);
                bar();
            }
            if constexpr (epi_is_lnfold(EPI)) HX_WAIT_VM(8);      // the tile-start statistics / bias pieces are older than the 8 newest ring pieces (a one-step tile has not retired them yet)
            epilogue_p<EPI, NI, epi_is_lnfold(EPI), 8, RD, false, DBG, epi_is_lnfold(EPI)>(p, acc, stg, M0 + wr * 128, N0 + wc * 64, lane);
        } else {
            // the same wait / DMA / barrier skeleton for a wave whose output block lies in the padding of a ragged edge tile
            for (int t = 0; t < nst; ++t, ++g) {
                HX_WAIT_VM(8);
                if (a1_valid) issue_a1_pending();
                bar(); bar();
                HX_WAIT_VM(8);
                bar(); bar();
                issue_a0(); issue_w(0);
                bar(); bar();
                HX_WAIT_VM(8);
                issue_w(1);
                advance();
                a1_valid = true;
                bar(); bar();
            }
        }
    }
    if (grp == 0) bar();                              // pair the skew barrier
    HX_WAIT_VM(0);
}

template <int EPI, int RD = 1>
__global__ __launch_bounds__(512) void gemm_pp256(GemmP p) { pp256_body<EPI, RD, false>(p); }
template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp256x3(GemmP p) { pp256_body<EPI, 1, true>(p); }
template <int EPI>
__global__ __launch_bounds__(512) void gemm_pq256(GemmP p) { pp256_body<EPI, (EPI == HIREST_EPI_BIAS_RESID2_LNSTATS ? HIREST_S2_RD : 1), false, 1>(p); }      // two-phase schedule (PH2); the two-array residual epilogue decodes its loads a pass late: look-ahead 2 (1.258 -> 1.206 ms on proj; 3: 1.27-1.30)      // two-phase schedule (PH2)
template <int EPI>
__global__ __launch_bounds__(512) void gemm_pq256x3(GemmP p) { pp256_body<EPI, 1, true, 1>(p); }
template <int EPI>      // the measurement variant of gemm_pq256: same schedule and arithmetic, with the hirest_gemm_debug_mode switches live
__global__ __launch_bounds__(512) void gemm_pq256_dbg(GemmP p) { pp256_body<EPI, (EPI == HIREST_EPI_BIAS_RESID2_LNSTATS ? HIREST_S2_RD : 1), false, 1, true>(p); }


template <int EPI, int RD = 1, bool X3 = false, int PH2 = 0, bool DBG = false>
int launch_pp256(GemmP p, hipStream_t s) {
    static_assert(!DBG || (PH2 && !X3), "the switch-carrying instantiation exists for gemm_pq256 only");
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = [] {
        if constexpr (DBG) return gemm_pq256_dbg<EPI>;
        else if constexpr (X3 && PH2) return gemm_pq256x3<EPI>;
        else if constexpr (X3) return gemm_pp256x3<EPI>;
        else if constexpr (PH2) return gemm_pq256<EPI>;
        else return gemm_pp256<EPI, RD>;
    }();
    constexpr int LDS = 2 * Q_STEP + 8 * p_stg_bytes(EPI);
    if (int e = hirest_configure(kern, LDS, cfg, &cus)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    int nslot = cus / 8; nslot = nslot < 1 ? 1 : nslot;
    const int per_xcd = p.ppx * p.nbn;
    if (nslot > per_xcd) nslot = per_xcd;
    hipLaunchKernelGGL(kern, dim3(8 * nslot), dim3(512), LDS, s, p);
    return hirest_launch_status();
}

template <int EPI, int WN, bool DBG, int RD = 1>
int launch_p256_impl(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = gemm_p256<EPI, WN, DBG, RD>;
    constexpr int NW = 512 / WN;
    constexpr int LDS = 2 * Q_STEP + NW * p_stg_bytes(EPI);
    if (int e = hirest_configure(kern, LDS, cfg, &cus)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    int nslot = cus / 8; nslot = nslot < 1 ? 1 : nslot;
    const int per_xcd = p.ppx * p.nbn;
    if (nslot > per_xcd) nslot = per_xcd;
    hipLaunchKernelGGL(kern, dim3(8 * nslot), dim3(64 * NW), LDS, s, p);
    return hirest_launch_status();
}

template <int EPI, int WN>
int launch_p256(GemmP p, hipStream_t s) {
    if constexpr (WN == 64 && (EPI == HIREST_EPI_BIAS_BF16 || EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_BIAS_RESID_F32)) {
        if (p.dbg) return launch_p256_impl<EPI, WN, true>(p, s);       // experiment kernel: the three tower epilogues only
    }
    p.dbg = 0;
    return launch_p256_impl<EPI, WN, false>(p, s);
}

template <int EPI>
int launch256q(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_t256q<EPI>;
    if (int e = hirest_configure(kern, 2 * Q_STEP, cfg)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    const int grid = 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * Q_STEP, s, p);
    return hirest_launch_status();
}

template <int EPI>
int launch256p(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_t256p<EPI>;
    if (int e = hirest_configure(kern, 4 * T_SLAB, cfg)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    const int grid = 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 4 * T_SLAB, s, p);
    return hirest_launch_status();
}

template <int EPI, int T_NST>
int launch256(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_t256<EPI, T_NST>;
    if (int e = hirest_configure(kern, T_NST * T_SLAB, cfg)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    const int grid = 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), T_NST * T_SLAB, s, p);
    return hirest_launch_status();
}

}  // namespace
std::atomic<int> g_gemm_dbg{0};
namespace {

template <int EPI, int WM>
int launch_t128x3_impl(const GemmP& p0, hipStream_t s) {
    static HirestDevCfg cfg;
    auto kern = gemm_t128x3<EPI, WM>;
    constexpr int LDS = 4 * (64 * WM + BN) * BK * 2;
    if (int e = hirest_configure(kern, LDS, cfg)) return e;
    GemmP p = p0;
    p.nbm = (p.M + 64 * WM - 1) / (64 * WM);
    p.ppx = (p.nbm + 7) / 8;
    p.flat = p.nbm < 32 ? 1 : 0;                                  // (same tiles, same arithmetic per tile: the mapping does not change a bit of the result)
    // Split-K (HIREST_EPI_BIAS_RESID_F32 with a scratch buffer in aux0): a 768-wide layer over 1 500 rows is 72 tiles for 256 CUs and the 3072-deep
    // one of them (output.dense) runs 96 steps per tile — S slices of the K range per tile fill the chip; the slices are added in a fixed order.
    p.ksplit = 1;
    if (EPI == HIREST_EPI_BIAS_RESID_F32 && p.flat && p.aux0 && !p.aux1) {
        const int ntile = p.nbm * p.nbn, nk = p.K / BK;
        for (int S = 4; S >= 2; --S)
            if (nk % S == 0 && nk / S >= 8 && ntile * S <= 256) { p.ksplit = S; break; }
        p.part = reinterpret_cast<float*>(p.aux0);
    }
    const int grid = p.flat ? 8 * ((p.nbm * p.nbn * p.ksplit + 7) / 8) : 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * WM), LDS, s, p);
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.M * (p.N / 4);
        int blocks = (int)((total + 255) / 256); blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.part, p.ksplit, p.bias, reinterpret_cast<float*>(p.out), p.ldo, p.M, p.N);
    }
    return hirest_launch_status();
}
// 192-row tiles (WM = 3) when they save rounds over the 256 CUs: cost of a tiling = rounds x tile height.  Taken only under HIREST_GEMM_X3_T128
// (the joint model's calls): the towers' small split-operand calls keep the 128 x 128 tiles they have always had.
static inline bool t128x3_tall(const GemmP& p, bool allow) {
    if (!allow) return false;
    const int64_t n2 = (int64_t)((p.M + 127) / 128) * p.nbn, n3 = (int64_t)((p.M + 191) / 192) * p.nbn;
    const int64_t c2 = ((n2 + 255) / 256) * 2, c3 = ((n3 + 255) / 256) * 3;
    return 4 * c3 <= 3 * c2;                                      // (a 6 % saving on paper measured 2 % slower at B = 32: take it from a quarter on)
}
template <int EPI>
int launch_t128x3(const GemmP& p, hipStream_t s, bool allow_tall = false) {
    return t128x3_tall(p, allow_tall) ? launch_t128x3_impl<EPI, 3>(p, s) : launch_t128x3_impl<EPI, 2>(p, s);
}
static inline bool x3_small(int64_t M, int64_t N) { return ((M + 255) / 256) * ((N + 255) / 256) < 256; }
std::atomic<int> g_force_kernel{0};   // 0 auto, 1 t128, 2 t256 with a 4-slot ring, 3 t256 with a 5-slot ring (tests / A-B timing)

// walk / knock-out switches of hirest_gemm_debug_mode that only gemm_pq256_dbg reads (bits 10-19)
static inline bool pq_switches(const GemmP& p) { return p.sched || p.stagger || p.epi_dbg; }

// LN-fold epilogues exist in the persistent kernels only
template <int EPI>
int launch_fused(const GemmP& p, hipStream_t s) {
    // (the persistent kernels take any M, N: a small problem just leaves CUs idle — the CLS-row GEMMs of a tower's last block)
    const bool big = p.M >= 64 && p.N >= 256;
    if (!big || !p.aux0 || !p.aux1) return !big ? HIREST_E_SHAPE : HIREST_E_BADARG;
    if ((EPI == HIREST_EPI_BIAS_RESID_LNSTATS_F32 || EPI == HIREST_EPI_BIAS_RESID2_LNSTATS) && p.N % 8 != 0) return HIREST_E_SHAPE;   // 16-B stores of the bf16 copy
    if (EPI == HIREST_EPI_BIAS_RESID2_LNSTATS && p.ldo % 8 != 0) return HIREST_E_SHAPE;
    GemmP q = p; q.dbg = 0;
    // default since round 4: the two-phase ping-pong kernel (pq256) on every shape — 1-3.3 % faster than p256 (K = 1408) / pp256 (K = 6144),
    // same bits; 6 / 8 select those for A/B
    if (g_force_kernel == 6) return launch_p256_impl<EPI, 64, false>(q, s);
    if (g_force_kernel == 8) return launch_pp256<EPI>(q, s);
    if (pq_switches(q)) return launch_pp256<EPI, 1, false, 1, true>(q, s);      // measurement variant (gemm_pq256_dbg)
    return launch_pp256<EPI, 1, false, 1>(q, s);
}


template <int EPI>
int launch(const GemmP& p, hipStream_t s) {
    const bool big = (int64_t)p.M * p.N >= (int64_t)2048 * 1024 && p.M >= 512 && p.N >= 256;
    // large problems: the two-phase ping-pong kernel (round 4; p256 / pp256 stay selectable: 6 / 8)
    constexpr bool has_dbg_inst = EPI == HIREST_EPI_BIAS_BF16 || EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_BIAS_RESID_F32;
    if (g_force_kernel == 0 && big && !(p.dbg && has_dbg_inst))
        return pq_switches(p) ? launch_pp256<EPI, 1, false, 1, true>(p, s) : launch_pp256<EPI, 1, false, 1>(p, s);
    if (g_force_kernel == 6 || (g_force_kernel == 0 && big)) return launch_p256<EPI, 64>(p, s);      // (timing-experiment bits exist in p256 only)
    if (g_force_kernel == 7) return launch_p256<EPI, 128>(p, s);
    if (g_force_kernel == 8) return launch_pp256<EPI>(p, s);
    if (g_force_kernel == 9) return pq_switches(p) ? launch_pp256<EPI, 1, false, 1, true>(p, s) : launch_pp256<EPI, 1, false, 1>(p, s);
    if (g_force_kernel == 5) return launch256q<EPI>(p, s);
    if (g_force_kernel == 4) return launch256p<EPI>(p, s);
    if (g_force_kernel == 2) return launch256<EPI, 4>(p, s);
    if (g_force_kernel == 3) return launch256<EPI, 5>(p, s);
    const int grid = 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(gemm_t128<EPI>, dim3(grid), dim3(256), 2 * STAGE_BYTES, s, p);
    return hirest_launch_status();
}

}  // namespace

extern "C" int hirest_gemm_debug_mode(int32_t bits) { g_gemm_dbg = bits; return 0; }

extern "C" int hirest_gemm_select_kernel(int32_t which) {
    // 9..17 were the 4-wave kernel gemm_w4 and its schedule experiments (round 2; 3-7 % slower than p256 / pp256 on every shape, retired in
    // round 3), 18..20 the two-workgroup kernel gemm_d2 (round 3; 5-34 % slower, retired in round 4): DESIGN 4.1c / 4.1d, git history.
    if (which < 0 || which > 9) return HIREST_E_BADARG;      // 9: the persistent ping-pong kernel with two phases per step (pq256)
    g_force_kernel = which;
    return 0;
}

// Which kernel instantiation hirest_gemm_bf16 launches for these arguments under the current hirest_gemm_select_kernel /
// hirest_gemm_debug_mode state, spelled the way rocprofv3 prints kernel names (template arguments included).  Mirrors
// launch() / launch_fused() above; pure host logic (no launch, usable without a GPU).  Profiles committed under profiles/
// are checked against it (tests/test_abi_and_host.py), so a profile of kernels the tower no longer runs cannot be quoted.
extern "C" int hirest_gemm_dispatch_name(const hirest_gemm_args* a, char* out, int32_t out_len) {
    if (!a || a->struct_size != sizeof(hirest_gemm_args) || !out || out_len < 48) return HIREST_E_BADARG;
    const int epi = a->epilogue, f = g_force_kernel;
    if (epi < 0 || epi > HIREST_EPI_BIAS_RESID2_LNSTATS) return HIREST_E_BADARG;
    if (a->flags & HIREST_GEMM_X3) {
        if (epi != HIREST_EPI_BIAS_F32 && epi != HIREST_EPI_BIAS_RESID_F32 && epi != HIREST_EPI_BIAS_GELU_SPLIT2) return HIREST_E_BADARG;
        if (((a->flags & HIREST_GEMM_X3_T128) || x3_small(a->M, a->N)) && (epi != HIREST_EPI_BIAS_GELU_SPLIT2 || (a->flags & HIREST_GEMM_X3_T128))) {
            GemmP q; q.M = a->M; q.nbn = (a->N + BN - 1) / BN;
            snprintf(out, out_len, t128x3_tall(q, (a->flags & HIREST_GEMM_X3_T128) != 0) ? "gemm_t128x3<%d, 3>" : "gemm_t128x3<%d, 2>", epi);
            return 0;
        }
        snprintf(out, out_len, g_force_kernel == 9 ? "gemm_pq256x3<%d>" : "gemm_pp256x3<%d>", epi);
        return 0;
    }
    if (epi == HIREST_EPI_BIAS_GELU_SPLIT2) return HIREST_E_BADARG;      // exists in the X3 form only
    const bool fused = (epi >= HIREST_EPI_BIAS_RESID_LNSTATS_F32 && epi <= HIREST_EPI_LNFOLD_GELU_BF16) || epi == HIREST_EPI_BIAS_RESID2_LNSTATS;
    const bool big = fused ? (a->M >= 64 && a->N >= 256) : ((int64_t)a->M * a->N >= (int64_t)2048 * 1024 && a->M >= 512 && a->N >= 256);
    const bool dbg_inst = !fused && (g_gemm_dbg & ~(512 | 3072 | 0xF000 | 0xF0000)) && (epi == HIREST_EPI_BIAS_BF16 || epi == HIREST_EPI_BIAS_GELU_BF16 || epi == HIREST_EPI_BIAS_RESID_F32);
    if (fused && !big) return HIREST_E_SHAPE;
    const bool sw = (g_gemm_dbg & (3072 | 0xF000 | 0xF0000)) != 0;      // stagger / epilogue knock-outs / walk switches: the measurement variant
    if (f == 9 || (fused && f != 6 && f != 8) || (!fused && f == 0 && big && !dbg_inst)) snprintf(out, out_len, sw ? "gemm_pq256_dbg<%d>" : "gemm_pq256<%d>", epi);
    else if (fused) {
        if (f == 8) snprintf(out, out_len, "gemm_pp256<%d, 1>", epi);
        else snprintf(out, out_len, "gemm_p256<%d, 64, false, 1>", epi);
    } else if (f == 6 || (f == 0 && big)) snprintf(out, out_len, "gemm_p256<%d, 64, %s, 1>", epi, dbg_inst ? "true" : "false");
    else if (f == 7) snprintf(out, out_len, "gemm_p256<%d, 128, false, 1>", epi);
    else if (f == 8) snprintf(out, out_len, "gemm_pp256<%d, 1>", epi);
    else if (f == 5) snprintf(out, out_len, "gemm_t256q<%d>", epi);
    else if (f == 4) snprintf(out, out_len, "gemm_t256p<%d>", epi);
    else if (f == 2) snprintf(out, out_len, "gemm_t256<%d, 4>", epi);
    else if (f == 3) snprintf(out, out_len, "gemm_t256<%d, 5>", epi);
    else snprintf(out, out_len, "gemm_t128<%d>", epi);
    return 0;
}

extern "C" int hirest_gemm_bf16(const hirest_gemm_args* a, void* stream) {
    if (!a || a->struct_size != sizeof(hirest_gemm_args) || !a->A || !a->W || !a->out) return HIREST_E_BADARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return HIREST_E_BADARG;
    if (a->flags & ~(HIREST_GEMM_REVERSE | HIREST_GEMM_X3 | HIREST_GEMM_X3_T128)) return HIREST_E_BADARG;
    if ((a->flags & HIREST_GEMM_X3_T128) && !(a->flags & HIREST_GEMM_X3)) return HIREST_E_BADARG;      // (a retired flag, e.g. round 4's K-blocked operands, must not be read as row-major)
    if (a->K % BK != 0 || a->K % T_BK != 0 || a->N % 4 != 0 || a->lda % 8 != 0 || a->ldw % 8 != 0) return HIREST_E_SHAPE;
    GemmP p;
    p.A = reinterpret_cast<const bf16_t*>(a->A); p.lda = a->lda;
    p.W = reinterpret_cast<const bf16_t*>(a->W); p.ldw = a->ldw;
    p.bias = a->bias; p.out = a->out; p.ldo = a->ldo;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.pos = a->pos; p.P = a->patches_per_frame;
    p.aux0 = a->aux0; p.aux1 = a->aux1;
    p.rev = ((a->flags & HIREST_GEMM_REVERSE) && !(g_gemm_dbg & 512)) ? 1 : 0;   // debug bit 9: ignore the direction flags (A/B)
    p.dbg = g_gemm_dbg & ~(512 | 3072 | 0xF000 | 0xF0000);
    p.sched = (g_gemm_dbg >> 16) & 15;     // bit 18: team walk of the persistent ping-pong kernels (few column tiles); bit 16: uneven XCD split; bit 17: the two-array residual epilogue loads hi / lo cached instead of streaming (A/B)
    p.stagger = (g_gemm_dbg >> 10) & 3;
    p.epi_dbg = (g_gemm_dbg >> 12) & 15;   // A/B experiment: start the CUs of an XCD 0..3 quarter tiles apart (bits 10-11 = mode)
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN;
    p.ppx = (p.nbm + 7) / 8;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HirestProfScope prof(HIREST_PROF_GEMM, a->epilogue, a->M, a->N, a->K, s);
    if (a->flags & HIREST_GEMM_X3) {                  // split-operand products: the ping-pong kernel's X3 form, fp32 outputs only
        // fewer 256 x 256 tiles than CUs: the 128 x 128 kernel (2 workgroups per CU)
        const bool small = x3_small(a->M, a->N) || (a->flags & HIREST_GEMM_X3_T128);
        const bool tall = (a->flags & HIREST_GEMM_X3_T128) != 0;
        if (small && a->epilogue == HIREST_EPI_BIAS_F32) return launch_t128x3<HIREST_EPI_BIAS_F32>(p, s, tall);
        if (small && a->epilogue == HIREST_EPI_BIAS_RESID_F32) return launch_t128x3<HIREST_EPI_BIAS_RESID_F32>(p, s, tall);
        if ((a->flags & HIREST_GEMM_X3_T128) && a->epilogue == HIREST_EPI_BIAS_GELU_SPLIT2) {      // (the towers keep the 256 x 256 kernel for this epilogue at every size)
            if (a->N % 32 != 0 || a->ldo < 2 * (int64_t)a->N || a->ldo % 8 != 0) return HIREST_E_SHAPE;
            return launch_t128x3<HIREST_EPI_BIAS_GELU_SPLIT2>(p, s, true);
        }
        switch (a->epilogue) {
            case HIREST_EPI_BIAS_F32:
                return g_force_kernel == 9 ? launch_pp256<HIREST_EPI_BIAS_F32, 1, true, 1>(p, s) : launch_pp256<HIREST_EPI_BIAS_F32, 1, true>(p, s);
            case HIREST_EPI_BIAS_RESID_F32:
                return g_force_kernel == 9 ? launch_pp256<HIREST_EPI_BIAS_RESID_F32, 1, true, 1>(p, s)
                                           : launch_pp256<HIREST_EPI_BIAS_RESID_F32, 1, true>(p, s);
            case HIREST_EPI_BIAS_GELU_SPLIT2:
                if (a->N % 32 != 0 || a->ldo < 2 * (int64_t)a->N || a->ldo % 8 != 0) return HIREST_E_SHAPE;
                return g_force_kernel == 9 ? launch_pp256<HIREST_EPI_BIAS_GELU_SPLIT2, 1, true, 1>(p, s)
                                           : launch_pp256<HIREST_EPI_BIAS_GELU_SPLIT2, 1, true>(p, s);
            default: return HIREST_E_BADARG;
        }
    }
    switch (a->epilogue) {
        case HIREST_EPI_BIAS_BF16: return launch<HIREST_EPI_BIAS_BF16>(p, s);
        case HIREST_EPI_BIAS_GELU_BF16: return launch<HIREST_EPI_BIAS_GELU_BF16>(p, s);
        case HIREST_EPI_BIAS_QGELU_BF16: return launch<HIREST_EPI_BIAS_QGELU_BF16>(p, s);
        case HIREST_EPI_BIAS_RESID_F32: return launch<HIREST_EPI_BIAS_RESID_F32>(p, s);
        case HIREST_EPI_BIAS_F32: return launch<HIREST_EPI_BIAS_F32>(p, s);
        case HIREST_EPI_PATCH_POS_F32:
            if (!a->pos || a->patches_per_frame <= 0) return HIREST_E_BADARG;
            return launch<HIREST_EPI_PATCH_POS_F32>(p, s);
        case HIREST_EPI_BIAS_RESID_LNSTATS_F32: return launch_fused<HIREST_EPI_BIAS_RESID_LNSTATS_F32>(p, s);
        case HIREST_EPI_BIAS_RESID2_LNSTATS: return launch_fused<HIREST_EPI_BIAS_RESID2_LNSTATS>(p, s);
        case HIREST_EPI_LNFOLD_BF16: return launch_fused<HIREST_EPI_LNFOLD_BF16>(p, s);
        case HIREST_EPI_LNFOLD_GELU_BF16: return launch_fused<HIREST_EPI_LNFOLD_GELU_BF16>(p, s);
        default: return HIREST_E_BADARG;     // (HIREST_EPI_BIAS_GELU_SPLIT2 without HIREST_GEMM_X3 included)
    }
}
