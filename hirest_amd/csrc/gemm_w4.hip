// Kernel "w4": the persistent 256x256 tile with FOUR waves, one per SIMD, each owning a 128x128 block of the tile
// (8x8 tiles of v_mfma_f32_16x16x32_bf16, 256 accumulator registers in AGPRs).  Same ring (2 x 64 KiB, 64-deep K steps,
// XOR-swizzled 128-B rows filled by LDS-DMA), same continuous DMA stream across tiles and same epilogues as p256
// (gemm.hip); what changes is who feeds the matrix pipes:
//   * p256 puts two waves of 128x64 on every SIMD.  Per 64-deep step the CU issues 8 x 24 = 192 ds_read_b128 and the two
//     waves of a SIMD take turns on its matrix pipe, meeting at one barrier per step.
//   * w4: 4 x 32 = 128 fragment reads per step (2/3), and every SIMD's pipe is fed by ONE in-order instruction stream in
//     which each memory instruction sits in the shadow of an MFMA (16 cycles each): 128 MFMA + 32 ds_read_b128 +
//     16 LDS-DMA pieces per wave and step.
// One step (iteration g, ring slot s = g & 1 holds step g, the other slot step g + 1):
//   P1  32 MFMA (k 0..31, rows 0..63)    | 16 fragment reads: k 32..63 of step g            -> lgkmcnt(0), BARRIER 1
//   P2  32 MFMA (k 0..31, rows 64..127)  | 8 DMA pieces: W rows of step g+2 -> slot s
//   P3  32 MFMA (k 32..63, rows 0..63)   | 8 DMA pieces: A rows of step g+2 -> slot s      -> vmcnt(16), BARRIER 2
//   P4  32 MFMA (k 32..63, rows 64..127) | 16 fragment reads: k 0..31 of step g+1 (other slot)
// WAR: after barrier 1 every wave has read everything it needs from slot s (its k 0..31 fragments were read in P4 of the
//      previous iteration), so the refill may start.
// RAW: the pieces of step g+1 were issued during P2/P3 of iteration g-1; "all but the 16 newest" (= step g+2's) have
//      landed for this wave at the vmcnt(16), and for every wave after barrier 2.  Loads complete in order, so older
//      epilogue stores or statistics loads still in flight only make the wait stricter.
// A refill therefore has between 1.0 and 1.5 steps to land (p256: exactly one, issued as a burst of 8).
//
// Ragged N edge (N = 1408 = 5.5 tiles, 4224 = 16.5): a last column tile with at most 128 valid columns would leave the
// two SIMDs of the right wave column idle.  Such a tile is split 4 x 1 instead: wave w takes rows 64 w .. 64 w + 63 of the
// tile and all 128 valid columns (4 x 8 MFMA tiles), so every SIMD stays busy and the tile takes half the time:
//   E1  16 MFMA | 12 fragment reads (k 32..63)   -> lgkmcnt(0), BARRIER 1
//   E2  32 MFMA | 16 DMA pieces (step g+2)       -> vmcnt(16), BARRIER 2
//   E3  16 MFMA | 12 fragment reads (k 0..31 of step g+1)
#include "gemm_shared.h"

namespace {

// FLAGS (A/B timing of the schedule; 0 = production): bit0 no LDS-DMA in the loop (results wrong: structural ceiling),
// bit1 all 16 pieces in P2 (one per 2 MFMAs) instead of W in P2 / A in P3 (one per 4), bit2 fragment reads one per 2 MFMAs
// across the whole phase instead of one per MFMA in its first half.
template <int EPI, int FLAGS>
__global__ __launch_bounds__(256) void gemm_w4(GemmP p) {
    constexpr int NI = 8;           // 16-column MFMA tiles per wave (and 8 16-row tiles)
    constexpr int PPW = 8;          // LDS-DMA pieces per operand per wave per step (32 pieces of 8 rows each / 4 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- this block's tile list (identical to p256: XCD = blockIdx % 8 owns a contiguous range of M panels)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nslot = gridDim.x >> 3;
    const int p_lo = xcd * p.ppx;
    int np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np;
    if (np <= 0) return;
    const bool panel_major = p.nbn <= 8;
    const int nunit = np * p.nbn;
    if (slot >= nunit) return;
    auto tile_origin = [&](int j, int& M0, int& N0) {
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

    // ---- LDS-DMA stream state (runs two K steps ahead of the MFMAs, across tile boundaries).  Pieces are fetched with
    // buffer_load ... lds through one resource descriptor per operand whose base is the tile's first row and whose size is
    // what is left of the matrix from there: rows past M (or N) read as out of range instead of being clamped per lane, so
    // the 16 per-lane offsets are computed ONCE for the whole kernel and a tile switch is scalar work only.
    const int nst = p.K / Q_BK;
    uint32_t a_off[PPW], w_off[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int row = (wave * PPW + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        a_off[q] = (uint32_t)(row * (int)p.lda + chunk * 8) * 2u;
        w_off[q] = (uint32_t)(row * (int)p.ldw + chunk * 8) * 2u;
    }
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    auto set_sources = [&](int M0, int N0) {
        const uint64_t a_left = (uint64_t)(p.M - M0) * (uint64_t)p.lda * 2u, w_left = (uint64_t)(p.N - N0) * (uint64_t)p.ldw * 2u;
        a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + (int64_t)M0 * p.lda), 0,
                                                 (int)(uint32_t)(a_left < 0xFFFFFFFFull ? a_left : 0xFFFFFFFFull), 0x00020000);
        w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (int64_t)N0 * p.ldw), 0,
                                                 (int)(uint32_t)(w_left < 0xFFFFFFFFull ? w_left : 0xFFFFFFFFull), 0x00020000);
    };
    int dma_j = slot, dma_k = 0, dma_g = 0;
    bool dma_live = true;
    auto stage_a = [&]() {
        char* buf = smem + (dma_g & 1) * Q_STEP + wave * (PPW * 1024);
#pragma unroll
        for (int q = 0; q < PPW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, a_off[q],
                                                     dma_k * (Q_BK * 2), 0, 0);
    };
    auto stage_w = [&]() {
        char* buf = smem + (dma_g & 1) * Q_STEP + Q_WOFF + wave * (PPW * 1024);
#pragma unroll
        for (int q = 0; q < PPW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, w_off[q],
                                                     dma_k * (Q_BK * 2), 0, 0);
    };
    auto stage_a_range = [&](int q0, int q1) {
        char* buf = smem + (dma_g & 1) * Q_STEP + wave * (PPW * 1024);
#pragma unroll
        for (int q = q0; q < q1; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, a_off[q],
                                                     dma_k * (Q_BK * 2), 0, 0);
    };
    auto stage_w_range = [&](int q0, int q1) {
        char* buf = smem + (dma_g & 1) * Q_STEP + Q_WOFF + wave * (PPW * 1024);
#pragma unroll
        for (int q = q0; q < q1; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, w_off[q],
                                                     dma_k * (Q_BK * 2), 0, 0);
    };
    auto advance = [&]() {   // wave-uniform; past the last tile the stream refetches its last step (never consumed)
        ++dma_g;
        if (dma_live && ++dma_k == nst) {
            dma_j += nslot;
            if (dma_j < nunit) { int m0, n0; tile_origin(dma_j, m0, n0); set_sources(m0, n0); dma_k = 0; }
            else { dma_live = false; dma_k = nst - 1; }
        }
    };

    // ---- fragments of v_mfma_f32_16x16x32_bf16: lane -> row (lane & 15), 8 consecutive k at 8 * (lane >> 4).
    // Addresses = (uniform: ring slot + operand + this wave's block) + (per lane: row * 128 + swizzled chunk): three VGPRs
    // serve every fragment read of both tile forms.
    const int frow = lane & 15, fsw = (frow >> 1) & 7, kg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) foff[h] = frow * (Q_BK * 2) + (((4 * h + kg) ^ fsw) << 4);
    const int a_blk = wr * 128 * (Q_BK * 2), w_blk = Q_WOFF + wc * 128 * (Q_BK * 2);               // 2 x 2 form
    const int a_blk_e = wave * 64 * (Q_BK * 2), w_blk_e = Q_WOFF;                                   // 4 x 1 form (edge tiles)

    auto read_frags = [&](const char* buf, int h, bf16x8 (&a)[8], bf16x8 (&w)[8]) {
#pragma unroll
        for (int n = 0; n < 8; ++n) w[n] = *reinterpret_cast<const bf16x8*>(buf + w_blk + n * 16 * 128 + foff[h]);
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const bf16x8*>(buf + a_blk + m * 16 * 128 + foff[h]);
    };
    auto read_w = [&](const char* buf, int h, bf16x8 (&w)[8]) {
#pragma unroll
        for (int n = 0; n < 8; ++n) w[n] = *reinterpret_cast<const bf16x8*>(buf + w_blk + n * 16 * 128 + foff[h]);
    };
    auto read_a = [&](const char* buf, int h, bf16x8 (&a)[8]) {
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const bf16x8*>(buf + a_blk + m * 16 * 128 + foff[h]);
    };
    // MFMAs i0 .. i1-1 of one 32-deep half-step in (row tile, column tile) order: i -> (i / 8, i % 8)
    auto mfma_span = [&](f32x4 (&acc)[8][NI], int i0, int i1, bf16x8 (&a)[8], bf16x8 (&w)[8]) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            acc[i >> 3][i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i & 7], a[i >> 3], acc[i >> 3][i & 7], 0, 0, 0);
    };
#define HX_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto mfma_rows = [&](f32x4 (&acc)[8][NI], int m0, int cnt, bf16x8 (&a)[8], bf16x8 (&w)[8]) {   // cnt row tiles x 8 column tiles
#pragma unroll
        for (int m = m0; m < m0 + cnt; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n], a[m], acc[m][n], 0, 0, 0);
    };

    auto interleave_reads = [&]() {      // 16 ds_read_b128 among 32 MFMAs
        if constexpr (FLAGS & 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    };

    // ---- prologue of the stream: steps 0 and 1 issued, step 0 landed everywhere
    {
        int m0, n0;
        tile_origin(slot, m0, n0);
        set_sources(m0, n0);
    }
    // (W before A, the order every loop below issues a step in: the counted waits of schedule H name pieces by position)
    stage_w(); stage_a(); advance();
    stage_w(); stage_a(); advance();
    HX_WAIT_VM(16);                      // step 0 landed (the 16 newest pieces are step 1's)
    __builtin_amdgcn_s_barrier();

    char* stg = smem + 2 * Q_STEP + wave * p_stg_bytes(EPI);
    int g = 0;   // global step index of the MFMA side: step g lives in ring slot g & 1
    for (int j = slot; j < nunit; j += nslot) {
        int M0, N0;
        tile_origin(j, M0, N0);
        const bool edge = p.N - N0 <= 128;                       // at most the left 128 columns exist: 4 x 1 split
        const int Mw = edge ? M0 + wave * 64 : M0 + wr * 128;   // this wave's block origin
        const int Nw = edge ? N0 : N0 + wc * 128;
        const bool active = Nw < p.N && Mw < p.M;
        if (active) {
            if constexpr (epi_is_lnfold(EPI)) {   // (mean, rstd) of this wave's rows (128 loaded): global -> LDS by DMA, lands under the K loop
                int r0 = Mw + 2 * lane;
                const int last = (p.M - 1) & ~1;   // the stats buffer is padded to an even number of rows
                r0 = r0 < last ? r0 : last;
                glds16(reinterpret_cast<const float*>(p.aux0) + 2 * (int64_t)r0, stg + P_STG);
            }
        }
        if (active && !edge) {
            bf16x8 A0[8], W0[8], A1[8], W1[8];      // k 0..31 and k 32..63 fragments of the current step
            f32x4 acc[8][NI];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            read_frags(smem + (g & 1) * Q_STEP, 0, A0, W0);
            if constexpr (FLAGS & 8) {
              // Schedule H: every memory instruction alone in the shadow of an MFMA, each operand's ring region released as
              // soon as its k 32..63 fragments are read, four barriers per step each between two MFMAs:
              //   k 0..31 :  8 reads W(k 32..63) | lgkm, B1 (W region of slot s free) | 5 DMA W(g+2) + 5 reads A(k 32..63)
              //              | 3 reads A | lgkm, B2 (A region free) | 3 DMA W + 2 DMA A(g+2)
              //   k 32..63:  vmcnt(18), B3 (W of step g+1 landed) | 8 reads W'(k 0..31, step g+1) | 5 DMA A
              //              | vmcnt(15), B4 (A of step g+1 landed) | 8 reads A' | 1 DMA A
              // vmcnt(18) = 8 A(g+1) + 8 W(g+2) + 2 A(g+2) still in flight allowed; vmcnt(15) = 8 W(g+2) + 7 A(g+2).
              for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                const char* nxt = smem + ((g + 1) & 1) * Q_STEP;
                __builtin_amdgcn_sched_barrier(0);
                // R1: MFMA 0..19 | 8 reads W1
                read_w(cur, 1, W1);
                mfma_span(acc, 0, 20, A0, W0);
                HX_SG(0x008, 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) { HX_SG(0x100, 1); HX_SG(0x008, 2); }
                HX_SG(0x008, 3);
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_LGKM0();
                mfma_span(acc, 20, 21, A0, W0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                                   // B1
                __builtin_amdgcn_sched_barrier(0);
                // R2: MFMA 21..49 | 5 DMA W + 8 reads A1
                // (program order interleaved: an LDS-DMA writes LDS, so the compiler never moves a ds_read across one)
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    stage_w_range(i, i + 1);
                    A1[i] = *reinterpret_cast<const bf16x8*>(cur + a_blk + i * 16 * 128 + foff[1]);
                }
#pragma unroll
                for (int i = 5; i < 8; ++i) A1[i] = *reinterpret_cast<const bf16x8*>(cur + a_blk + i * 16 * 128 + foff[1]);
                mfma_span(acc, 21, 50, A0, W0);
#pragma unroll
                for (int i = 0; i < 5; ++i) { HX_SG(0x008, 1); HX_SG(0x020, 1); HX_SG(0x008, 2); HX_SG(0x100, 1); }
#pragma unroll
                for (int i = 0; i < 3; ++i) { HX_SG(0x008, 2); HX_SG(0x100, 1); }
                HX_SG(0x008, 8);
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_LGKM0();
                mfma_span(acc, 50, 51, A0, W0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                                   // B2
                __builtin_amdgcn_sched_barrier(0);
                // R3: MFMA 51..63 | 3 DMA W + 2 DMA A
                stage_w_range(5, 8);
                stage_a_range(0, 2);
                mfma_span(acc, 51, 64, A0, W0);
#pragma unroll
                for (int i = 0; i < 5; ++i) { HX_SG(0x008, 2); HX_SG(0x020, 1); }
                HX_SG(0x008, 3);
                __builtin_amdgcn_sched_barrier(0);
                // R4: MFMA 64..66 (k 32..63)
                mfma_span(acc, 0, 3, A1, W1);
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_VM(18);
                mfma_span(acc, 3, 4, A1, W1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                                   // B3
                __builtin_amdgcn_sched_barrier(0);
                // R5: MFMA 68..103 | 8 reads W0 (step g+1) + 5 DMA A
                read_w(nxt, 0, W0);
                stage_a_range(2, 7);
                mfma_span(acc, 4, 40, A1, W1);
#pragma unroll
                for (int i = 0; i < 8; ++i) { HX_SG(0x008, 2); HX_SG(0x100, 1); }
#pragma unroll
                for (int i = 0; i < 5; ++i) { HX_SG(0x008, 2); HX_SG(0x020, 1); }
                HX_SG(0x008, 10);
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_VM(15);
                mfma_span(acc, 40, 41, A1, W1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                                   // B4
                __builtin_amdgcn_sched_barrier(0);
                // R6: MFMA 105..127 | 8 reads A0 (step g+1) + 1 DMA A
                read_a(nxt, 0, A0);
                stage_a_range(7, 8);
                mfma_span(acc, 41, 64, A1, W1);
#pragma unroll
                for (int i = 0; i < 8; ++i) { HX_SG(0x008, 2); HX_SG(0x100, 1); }
                HX_SG(0x008, 1); HX_SG(0x020, 1); HX_SG(0x008, 6);
                __builtin_amdgcn_sched_barrier(0);
                advance();
              }
            } else
            for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                const char* nxt = smem + ((g + 1) & 1) * Q_STEP;
                // ---- P1
                __builtin_amdgcn_sched_barrier(0);
                read_frags(cur, 1, A1, W1);
                mfma_rows(acc, 0, 4, A0, W0);
                // reads one per MFMA in the first half of the phase: the last one has 16 MFMAs (256 cycles) to return before
                // the lgkmcnt(0) below, so the wait costs nothing
                interleave_reads();
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- P2
                if constexpr (!(FLAGS & 1)) { stage_w(); if constexpr (FLAGS & 2) stage_a(); }
                mfma_rows(acc, 4, 4, A0, W0);
                if constexpr (FLAGS & 2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- P3
                if constexpr (!(FLAGS & 3)) stage_a();
                mfma_rows(acc, 0, 4, A1, W1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(FLAGS & 1)) HX_WAIT_VM(16);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- P4 (at the end of a tile these are the next tile's first fragments: reloaded after the epilogue)
                read_frags(nxt, 0, A0, W0);
                mfma_rows(acc, 4, 4, A1, W1);
                interleave_reads();                               // front-loaded as in P1: P1's first MFMA needs these
                __builtin_amdgcn_sched_barrier(0);
                advance();
            }
            epilogue_p<EPI, NI, epi_is_lnfold(EPI)>(p, acc, stg, Mw, Nw, lane);
        } else if (active) {
            // ---- ragged N edge tile, 4 x 1 split: rows 64 * wave .. + 63, the 128 valid columns
            auto read_edge = [&](const char* buf, int h, bf16x8 (&a)[8], bf16x8 (&w)[8]) {
#pragma unroll
                for (int n = 0; n < 8; ++n) w[n] = *reinterpret_cast<const bf16x8*>(buf + w_blk_e + n * 16 * 128 + foff[h]);
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const bf16x8*>(buf + a_blk_e + m * 16 * 128 + foff[h]);
            };
            bf16x8 A0[8], W0[8], A1[8], W1[8];                                 // (A: entries 0..3 only)
            f32x4 acc[8][NI];                                                  // rows 4..7 are never touched
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto mfma_pair = [&](int m0, bf16x8 (&a)[8], bf16x8 (&w)[8]) { mfma_rows(acc, m0, 2, a, w); };   // 16 MFMAs
            auto interleave12 = [&]() {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            };
            read_edge(smem + (g & 1) * Q_STEP, 0, A0, W0);
            for (int t = 0; t < nst; ++t, ++g) {
                const char* cur = smem + (g & 1) * Q_STEP;
                const char* nxt = smem + ((g + 1) & 1) * Q_STEP;
                // ---- E1
                __builtin_amdgcn_sched_barrier(0);
                read_edge(cur, 1, A1, W1);
                mfma_pair(0, A0, W0);
                interleave12();
                __builtin_amdgcn_sched_barrier(0);
                HX_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- E2
                if constexpr (!(FLAGS & 1)) { stage_w(); stage_a(); }
                mfma_pair(2, A0, W0);
                mfma_pair(0, A1, W1);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(FLAGS & 1)) HX_WAIT_VM(16);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- E3
                read_edge(nxt, 0, A0, W0);
                mfma_pair(2, A1, W1);
                interleave12();
                __builtin_amdgcn_sched_barrier(0);
                advance();
            }
            epilogue_p<EPI, NI, epi_is_lnfold(EPI), 4>(p, acc, stg, Mw, Nw, lane);
        } else {
            // the same barrier / DMA skeleton for a wave whose block lies outside the matrix (M edge): as many barriers per
            // step as the active waves of this tile execute (4 in schedule H's 2 x 2 form, 2 otherwise), W pieces after the
            // barrier that releases the W region, A pieces after the one that releases the A region
            const bool four = (FLAGS & 8) && !edge;
            for (int t = 0; t < nst; ++t, ++g) {
                __builtin_amdgcn_s_barrier();
                if constexpr (!(FLAGS & 1)) stage_w();
                if (four) __builtin_amdgcn_s_barrier();
                if constexpr (!(FLAGS & 1)) stage_a();
                HX_WAIT_VM(16);
                __builtin_amdgcn_s_barrier();
                if (four) __builtin_amdgcn_s_barrier();
                advance();
            }
        }
    }
    HX_WAIT_VM(0);
}

template <int EPI, int FLAGS = 0>
int launch_w4_impl(GemmP p, hipStream_t s) {
    static HirestDevCfg cfg;
    int cus = 0;
    auto kern = gemm_w4<EPI, FLAGS>;
    constexpr int LDS = 2 * Q_STEP + 4 * p_stg_bytes(EPI);
    if (int e = hirest_configure(kern, LDS, cfg, &cus)) return e;
    p.nbm = (p.M + T_BM - 1) / T_BM; p.nbn = (p.N + T_BN - 1) / T_BN;
    p.ppx = (p.nbm + 7) / 8;
    int nslot = cus / 8; nslot = nslot < 1 ? 1 : nslot;
    const int per_xcd = p.ppx * p.nbn;
    if (nslot > per_xcd) nslot = per_xcd;
    p.dbg = 0;
    hipLaunchKernelGGL(kern, dim3(8 * nslot), dim3(256), LDS, s, p);
    return hirest_launch_status();
}

}  // namespace

// Entry used by gemm.hip's dispatch (same GemmP, same preconditions as the other 256x256 kernels).
int hirest_launch_w4(int epi, const void* gemm_p, hipStream_t s, int flags) {
    const GemmP& p = *reinterpret_cast<const GemmP*>(gemm_p);
    if (flags && epi == HIREST_EPI_BIAS_BF16) {        // schedule experiments: plain epilogue only
        switch (flags) {
            case 1: return launch_w4_impl<HIREST_EPI_BIAS_BF16, 1>(p, s);
            case 2: return launch_w4_impl<HIREST_EPI_BIAS_BF16, 2>(p, s);
            case 4: return launch_w4_impl<HIREST_EPI_BIAS_BF16, 4>(p, s);
            case 6: return launch_w4_impl<HIREST_EPI_BIAS_BF16, 6>(p, s);
            default: break;
        }
    }
    if (flags == 8) {                                   // schedule H (gemm_w4<EPI, 8>): every tower epilogue
        switch (epi) {
            case HIREST_EPI_BIAS_BF16: return launch_w4_impl<HIREST_EPI_BIAS_BF16, 8>(p, s);
            case HIREST_EPI_BIAS_GELU_BF16: return launch_w4_impl<HIREST_EPI_BIAS_GELU_BF16, 8>(p, s);
            case HIREST_EPI_BIAS_RESID_F32: return launch_w4_impl<HIREST_EPI_BIAS_RESID_F32, 8>(p, s);
            case HIREST_EPI_BIAS_F32: return launch_w4_impl<HIREST_EPI_BIAS_F32, 8>(p, s);
            case HIREST_EPI_BIAS_RESID_LNSTATS_F32: return launch_w4_impl<HIREST_EPI_BIAS_RESID_LNSTATS_F32, 8>(p, s);
            case HIREST_EPI_LNFOLD_BF16: return launch_w4_impl<HIREST_EPI_LNFOLD_BF16, 8>(p, s);
            case HIREST_EPI_LNFOLD_GELU_BF16: return launch_w4_impl<HIREST_EPI_LNFOLD_GELU_BF16, 8>(p, s);
            default: return HIREST_E_BADARG;
        }
    }
    switch (epi) {
        case HIREST_EPI_BIAS_BF16: return launch_w4_impl<HIREST_EPI_BIAS_BF16>(p, s);
        case HIREST_EPI_BIAS_GELU_BF16: return launch_w4_impl<HIREST_EPI_BIAS_GELU_BF16>(p, s);
        case HIREST_EPI_BIAS_RESID_F32: return launch_w4_impl<HIREST_EPI_BIAS_RESID_F32>(p, s);
        case HIREST_EPI_BIAS_F32: return launch_w4_impl<HIREST_EPI_BIAS_F32>(p, s);
        case HIREST_EPI_BIAS_RESID_LNSTATS_F32: return launch_w4_impl<HIREST_EPI_BIAS_RESID_LNSTATS_F32>(p, s);
        case HIREST_EPI_LNFOLD_BF16: return launch_w4_impl<HIREST_EPI_LNFOLD_BF16>(p, s);
        case HIREST_EPI_LNFOLD_GELU_BF16: return launch_w4_impl<HIREST_EPI_LNFOLD_GELU_BF16>(p, s);
        default: return HIREST_E_BADARG;
    }
}
