// Kernel "d2": TWO independent persistent workgroups per CU, each 4 waves (one per SIMD) walking 256x128 tiles, so that one
// workgroup's epilogue (residual read, fp32 store, bf16 copy, row sums, GELU: HBM / VALU work with no MFMA issue) runs under the
// other workgroup's K loop.  The hardware interleaves the two instruction streams of a SIMD; there is no hand-written
// cross-group synchronisation.  (in p256 / pp256 the matrix pipes idle for 1.33 ms of every 12.67-ms layer
// while all eight waves of the CU's only workgroup run their epilogue together.)
//
// What a workgroup owns (80 KiB of the CU's 160 KiB of LDS, 256 of the 512 registers of every SIMD lane):
//   * tile 256 (M) x 128 (N), waves 2 x 2, each 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulators) — the same
//     per-wave block as p256, so the epilogues of gemm_shared.h are used unchanged;
//   * a 3-slot ring of 32-deep K steps: A 256 rows + W 128 rows of 64 B = 24 KiB per slot, filled by LDS-DMA in 1-KiB pieces
//     of 16 rows x 64 B (4 A pieces + 2 W pieces per wave and step).  Row image [row][4 x 16-B chunk], chunk' = chunk ^
//     (row & 8 ? 3 : 0), applied to the DMA's per-lane source address and to the ds_read_b128 fragment address.  A ds_read_b128
//     is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md,
//     LDS table) — i.e. rows 0-3 and 12-15 of one k group with rows 4-11 of its neighbour: with 64-B rows the 16-B slot of
//     the 256-B bank row is 4 (row & 3) + chunk', and the four rows that share (row & 3) get chunks kg, kg^1, kg^1^3, kg^3:
//     16 distinct slots per group;
//   * 2 KiB of epilogue staging per wave behind the ring; the LN-fold consumers take their row statistics through registers
//     (epilogue_p<..., SREG>), there is no LDS left for them.
// Price: a 256x128 tile pulls (256 + 128) / (256 * 128) operand rows per output element through L2 -> LDS, 1.5x the 256x256
// tile's, in 64-B instead of 128-B row segments, and there are two barriers per 64 deep instead of one.
//
// One iteration = one 32-deep step g (fragments of step g are in registers, read during iteration g - 1):
//     vmcnt(6) lgkmcnt(0) | BARRIER | DMA of step g+3 -> slot g % 3 | 12 fragment reads of step g+1 | 32 MFMA of step g
//   RAW  step g+1's pieces were issued in iteration g-2; the only newer pieces of this wave at the wait are step g+2's six, so
//        vmcnt(6) retires them, and the barrier publishes every wave's pieces before anyone reads step g+1.
//   WAR  slot g % 3 held step g, whose fragments every wave has received (lgkmcnt(0)) before the barrier; the refill is issued
//        after it.  A refill has two iterations (64 deep of MFMAs) to land, like p256's one 64-deep step.
// The DMA stream is continuous across tiles (the first three steps of the next tile land under the last three of this one and
// under the epilogue).
//
// De-phasing: the two workgroups of a CU are started half a K loop apart (the later half of the grid sleeps first); from then on
// the offset is neutrally stable — whoever runs alone during the other's epilogue gains exactly what it loses when it pays its own.
#include "gemm_shared.h"

namespace {

constexpr int D_BM = 256, D_BN = 128, D_BK = 32;
constexpr int D_WOFF = D_BM * D_BK * 2;                 // 16 KiB: W rows start here inside a slot
constexpr int D_STEP = (D_BM + D_BN) * D_BK * 2;        // 24 KiB
constexpr int D_NSLOT = 3;
constexpr int D_LDS = D_NSLOT * D_STEP + 4 * P_STG;     // 80 KiB

// FLAGS (A/B timing; 0 = production): bit0 no LDS-DMA in the loop (results wrong: what the K loop costs without its refill);
// bit1 the operands are ADDRESSED as if stored K-blocked, [K/32][rows][32] (results wrong, same bytes): every 1-KiB piece is then
// 8 whole 128-B lines instead of 16 half lines — what a blocked operand layout would buy
template <int EPI, int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_d2(GemmP p) {
    constexpr int NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- tile list: XCD (bid & 7) owns a contiguous range of M panels, its 2 x 32 workgroups take every nslot-th tile
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nslot = gridDim.x >> 3;
    int p_lo, np;
    xcd_panels(p, xcd, p_lo, np);
    if (np <= 0) return;
    const bool panel_major = p.nbn <= 16;               // N = 1408: 11 column tiles, a round of 64 = 5.8 panels x 11 columns
    const int nunit = np * p.nbn;
    if (slot >= nunit) return;
    auto tile_origin = [&](int j, int& M0, int& N0) {
        if (p.rev) j = nunit - 1 - j;
        if (panel_major) {
            const int mt_i = j / p.nbn;
            M0 = (p_lo + mt_i) * D_BM; N0 = (j - mt_i * p.nbn) * D_BN;
            return;
        }
        const int g_ = j / (GROUP_M * p.nbn);           // groups of 8 panels, column-major inside: a round = 8 panels x 8 columns
        const int r = j - g_ * GROUP_M * p.nbn;
        int gcount = np - g_ * GROUP_M; gcount = gcount > GROUP_M ? GROUP_M : gcount;
        const int nt_i = r / gcount, mt_i = g_ * GROUP_M + (r - nt_i * gcount);
        M0 = (p_lo + mt_i) * D_BM; N0 = nt_i * D_BN;
    };

    // ---- LDS-DMA stream (up to three steps ahead of the MFMAs, across tile boundaries)
    const int nst = p.K / D_BK;
    uint32_t a_off[4], w_off[2];
    const char* a_base; const char* w_base;
    auto set_sources = [&](int M0, int N0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (wave * 4 + q) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row & 8) ? 3 : 0);
            int ra = p.M - 1 - M0; ra = row < ra ? row : ra;
            a_off[q] = (uint32_t)(ra * (int)p.lda + chunk * 8) * 2u;
            if constexpr (FLAGS & 2) a_off[q] = (uint32_t)(ra * 32 + chunk * 8) * 2u;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (wave * 2 + q) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row & 8) ? 3 : 0);
            int rn = p.N - 1 - N0; rn = row < rn ? row : rn;
            w_off[q] = (uint32_t)(rn * (int)p.ldw + chunk * 8) * 2u;
            if constexpr (FLAGS & 2) w_off[q] = (uint32_t)(rn * 32 + chunk * 8) * 2u;
        }
        a_base = reinterpret_cast<const char*>(p.A + (int64_t)M0 * p.lda);
        w_base = reinterpret_cast<const char*>(p.W + (int64_t)N0 * p.ldw);
        if constexpr (FLAGS & 2) {
            a_base = reinterpret_cast<const char*>(p.A + (int64_t)M0 * 32);
            w_base = reinterpret_cast<const char*>(p.W + (int64_t)N0 * 32);
        }
    };
    int dma_j = slot, dma_k = 0, dma_slot = 0;
    bool dma_live = true;
    auto stage = [&]() {
        if constexpr (FLAGS & 1) return;
        char* buf = smem + dma_slot * D_STEP;
        const char* ab = a_base + (int64_t)dma_k * ((FLAGS & 2) ? (int64_t)p.M * 64 : (int64_t)(D_BK * 2));
        const char* wb = w_base + (int64_t)dma_k * ((FLAGS & 2) ? (int64_t)p.N * 64 : (int64_t)(D_BK * 2));
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(ab + a_off[q], buf + (wave * 4 + q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(wb + w_off[q], buf + D_WOFF + (wave * 2 + q) * 1024);
    };
    auto advance = [&]() {   // wave-uniform; past the last tile the stream refetches its last step (never consumed)
        dma_slot = dma_slot == D_NSLOT - 1 ? 0 : dma_slot + 1;
        if (dma_live && ++dma_k == nst) {
            dma_j += nslot;
            if (dma_j < nunit) { int m0, n0; tile_origin(dma_j, m0, n0); set_sources(m0, n0); dma_k = 0; }
            else { dma_live = false; dma_k = nst - 1; }
        }
    };

    // ---- fragments of v_mfma_f32_16x16x32_bf16: lane -> row (lane & 15), the 8 k at 8 * (lane >> 4): one 16-B chunk per 32-deep step
    const int frow = lane & 15, kg = lane >> 4;
    const int koff = (kg ^ ((frow & 8) ? 3 : 0)) << 4;
    const int a_frag = (wr * 128 + frow) * (D_BK * 2) + koff;
    const int w_frag = D_WOFF + (wc * 64 + frow) * (D_BK * 2) + koff;
    struct Frags { bf16x8 w[NI]; bf16x8 a[8]; };
    Frags f0, f1;
    auto load_frags = [&](const char* buf, Frags& f) {
#pragma unroll
        for (int n = 0; n < NI; ++n) f.w[n] = *reinterpret_cast<const bf16x8*>(buf + w_frag + n * 16 * (D_BK * 2));
#pragma unroll
        for (int m = 0; m < 8; ++m) f.a[m] = *reinterpret_cast<const bf16x8*>(buf + a_frag + m * 16 * (D_BK * 2));
    };
    f32x4 acc[8][NI];
    auto mfma_step = [&](Frags& fc) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc.w[n], fc.a[m], acc[m][n], 0, 0, 0);
    };

    // ---- de-phase the two workgroups of a CU (p.stagger: 1 = the later half of the grid, 2 = odd slots, 3 = none)
    {
        const bool late = p.stagger == 3 ? false : p.stagger == 2 ? (slot & 1) : (2 * slot >= nslot);
        if (late) {
            const int naps = (nst + 7) / 8;              // ~half a K loop: a 32-deep step is ~1000 cycles, s_sleep 127 ~ 8000
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- prologue of the stream: steps 0, 1, 2
    {
        int m0, n0;
        tile_origin(slot, m0, n0);
        set_sources(m0, n0);
    }
    stage(); advance();
    stage(); advance();
    stage(); advance();
    if constexpr (!(FLAGS & 1)) HX_WAIT_VM(12);          // step 0 landed (steps 1, 2 may still fly)
    __builtin_amdgcn_s_barrier();

    char* stg = smem + D_NSLOT * D_STEP + wave * P_STG;
    int rs = 0;                                          // ring slot of the MFMA side's current step
    auto next_slot = [](int s) { return s == D_NSLOT - 1 ? 0 : s + 1; };
    for (int j = slot; j < nunit; j += nslot) {
        int M0, N0;
        tile_origin(j, M0, N0);
        const bool active = (N0 + wc * 64 < p.N) && (M0 + wr * 128 < p.M);
        if (active) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            load_frags(smem + rs * D_STEP, f0);
            auto iteration = [&](Frags& fc, Frags& fn) {
                const int rn = next_slot(rs);
                if constexpr (!(FLAGS & 1)) HX_WAIT_VM(6);
                HX_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                stage();                                  // step g + 3 -> slot rs (its fragments are in registers everywhere)
                load_frags(smem + rn * D_STEP, fn);       // (last step of a tile: the next tile's step 0 — reloaded after the epilogue)
                mfma_step(fc);
                if constexpr (!(FLAGS & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                advance();
                rs = rn;
            };
            for (int t = 0; t < nst; t += 2) {            // nst = K / 32 is even (K % 64 == 0)
                iteration(f0, f1);
                iteration(f1, f0);
            }
            epilogue_p<EPI, NI, false, 8, 1, true>(p, acc, stg, M0 + wr * 128, N0 + wc * 64, lane);
        } else {
            for (int t = 0; t < nst; ++t) {
                if constexpr (!(FLAGS & 1)) HX_WAIT_VM(6);
                __builtin_amdgcn_s_barrier();
                stage();
                advance();
                rs = next_slot(rs);
            }
        }
    }
    HX_WAIT_VM(0);
}

template <int EPI, int FLAGS = 0>
int launch_d2_impl(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = gemm_d2<EPI, FLAGS>;
    if (int e = hirest_configure(kern, D_LDS, cfg, &cus)) return e;
    p.nbm = (p.M + D_BM - 1) / D_BM; p.nbn = (p.N + D_BN - 1) / D_BN;
    p.ppx = (p.nbm + 7) / 8;
    int nslot = 2 * cus / 8; nslot = nslot < 1 ? 1 : nslot;      // two workgroups per CU
    const int per_xcd = p.ppx * p.nbn;
    if (nslot > per_xcd) nslot = per_xcd;
    p.dbg = 0;
    hipLaunchKernelGGL(kern, dim3(8 * nslot), dim3(256), D_LDS, s, p);
    return hirest_launch_status();
}

}  // namespace

// Entry used by gemm.hip's dispatch (same GemmP, same preconditions as the other persistent kernels).
// flags: 1 = the no-DMA timing experiment (plain bf16 epilogue only).  GemmP::stagger selects the de-phasing (see the kernel).
int hirest_launch_d2(int epi, const void* gemm_p, hipStream_t s, int flags) {
    const GemmP& p = *reinterpret_cast<const GemmP*>(gemm_p);
    if (flags == 1 && epi == HIREST_EPI_BIAS_BF16) return launch_d2_impl<HIREST_EPI_BIAS_BF16, 1>(p, s);
    if (flags == 2 && epi == HIREST_EPI_BIAS_BF16) return launch_d2_impl<HIREST_EPI_BIAS_BF16, 2>(p, s);
    if (flags == 2 && epi == HIREST_EPI_BIAS_RESID_LNSTATS_F32) return launch_d2_impl<HIREST_EPI_BIAS_RESID_LNSTATS_F32, 2>(p, s);
    switch (epi) {
        case HIREST_EPI_BIAS_BF16: return launch_d2_impl<HIREST_EPI_BIAS_BF16>(p, s);
        case HIREST_EPI_BIAS_GELU_BF16: return launch_d2_impl<HIREST_EPI_BIAS_GELU_BF16>(p, s);
        case HIREST_EPI_BIAS_RESID_F32: return launch_d2_impl<HIREST_EPI_BIAS_RESID_F32>(p, s);
        case HIREST_EPI_BIAS_F32: return launch_d2_impl<HIREST_EPI_BIAS_F32>(p, s);
        case HIREST_EPI_BIAS_RESID_LNSTATS_F32: return launch_d2_impl<HIREST_EPI_BIAS_RESID_LNSTATS_F32>(p, s);
        case HIREST_EPI_LNFOLD_BF16: return launch_d2_impl<HIREST_EPI_LNFOLD_BF16>(p, s);
        case HIREST_EPI_LNFOLD_GELU_BF16: return launch_d2_impl<HIREST_EPI_LNFOLD_GELU_BF16>(p, s);
        default: return HIREST_E_BADARG;
    }
}
