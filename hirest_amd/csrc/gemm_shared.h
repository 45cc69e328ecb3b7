// Shared pieces of the 256x256 persistent GEMM kernels (gemm.hip: p256 / pp256, gemm_w4.hip: w4): parameter block, ring geometry,
// wait macros and the epilogues that turn a wave's accumulator block into whole-line HBM traffic.
#pragma once
#include "common.h"
#include "profile.h"

namespace {

struct GemmP {
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    int M, N, K;
    const float* pos; int P;
    int rev;                  // walk the tile list backwards (HIREST_GEMM_REVERSE)
    void* aux0; void* aux1;   // LN-fold epilogues (see hirest_hip.h): producer = bf16 copy / row partials, consumer = row stats / column sums
    int nbm, nbn, ppx;   // tile counts, M-panels per XCD
    int epi_dbg;         // timing experiment (hirest_gemm_debug_mode bits 12-15): LN-statistics epilogue without its bit0 residual read, bit1 f32 store, bit2 bf16 copy, bit3 row sums
    int stagger;         // timing experiment (hirest_gemm_debug_mode bits 10-11): staggered start of the CUs
    int sched;           // A/B switches of the tile schedule (hirest_gemm_debug_mode bit 16): uneven XCD split (ceil(nbm / 8) panels each)
    int dbg;             // timing experiments only (hirest_gemm_debug_mode): bit0 skip loop DMA, bit1 skip loop barrier+waits
};

constexpr int GROUP_M = 8;                        // M-panels walked together inside one XCD
// M panels of XCD x: an even split, [x nbm / 8, (x + 1) nbm / 8) (1028 panels = 4 x 129 + 4 x 128, not 7 x 129 + 125)
// DBG (here and in the epilogues below): the A/B and knock-out switches of hirest_gemm_debug_mode are read only by the *_dbg kernel
// instantiations; in the production code objects they fold to constants (a run-time branch on them cost gemm_pq256<10> a spilled register and
// 1 - 3 % of fc2 / proj in round 5).
template <bool DBG = false>
__device__ __forceinline__ void xcd_panels(const GemmP& p, int xcd, int& p_lo, int& np) {
    if (DBG && (p.sched & 1)) { p_lo = xcd * p.ppx; np = p.nbm - p_lo; np = np > p.ppx ? p.ppx : np; return; }   // A/B: the old split
    p_lo = (int)(((long long)xcd * p.nbm) >> 3);
    np = (int)(((long long)(xcd + 1) * p.nbm) >> 3) - p_lo;
}
constexpr int T_BM = 256, T_BN = 256;
#define HX_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define HX_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// 64-deep K steps through a 2-slot ring: A 256 rows + W 256 rows, 128 B each, chunk' = chunk ^ ((row>>1)&7)
constexpr int Q_BK = 64;
constexpr int Q_STEP = (T_BM + T_BN) * Q_BK * 2;   // 64 KiB
constexpr int Q_WOFF = T_BM * Q_BK * 2;

#ifndef HIREST_X3_GELU_POLY
#define HIREST_X3_GELU_POLY 1
#endif
constexpr int P_STG = 2048;    // epilogue staging bytes per wave: 16 rows x 128 B
constexpr bool epi_is_lnfold(int epi) { return epi == HIREST_EPI_LNFOLD_BF16 || epi == HIREST_EPI_LNFOLD_GELU_BF16; }
// the LN-fold consumers keep the (mean, rstd) pairs of the wave's 128 rows behind their staging area
constexpr int p_stg_bytes(int epi) { return epi_is_lnfold(epi) ? P_STG + 1024 : P_STG; }

// Epilogue of p256: accumulators are 16x16 MFMA tiles, acc[mi][ni][e] = C[mi*16 + (lane&15)][ni*16 + 4*(lane>>4) + e]
// (operands swapped, so a lane owns 4 consecutive columns of one row).  16 rows at a time go through a wave-private
// XOR-swizzled staging area so that HBM sees whole 128-B lines.
// Compiler-only ordering point for a wave-private LDS staging area: the LDS executes one wave's DS instructions in issue
// order (a ds_read issued after a ds_write of the same bytes returns the new data, a ds_write issued after a ds_read
// cannot overtake it), so no s_waitcnt is needed between the staging writes and the read-back — a wavefront-scope fence
// would drain lgkmcnt twice per 16-row pass.
#define HX_LDS_ORDER() asm volatile("" ::: "memory")

// Epilogue of HIREST_EPI_BIAS_RESID_LNSTATS_F32 (64-column wave tiles): x += acc + bias as in the plain residual form, plus
//   * the bf16 copy of the new rows (next GEMM's A operand): both 32-column halves of a pass are regrouped with one DPP
//     exchange per value so that every lane stores 16 B and a row's 64 columns leave as one full 128-B line;
//   * (sum, sum of squares) of the ROUNDED values per row and 64-column group -> aux1 [M, ceil(N/64), 2]
//     (hirest_ln_stats_finalize folds the groups into (mean, rstd)).
// The residual operands of pass mi + 1 are requested before pass mi is processed.
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {   // over the 8 lanes lane & ~7 .. | 7: quad xor 1, xor 2, then the mirrored quad
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    return v;
}
// `jc` = first 16-column MFMA tile of the 64-column group this call handles (a 128-column wave tile calls it twice), Nw its
// first column.
// NM = 16-row tiles of the block that hold results (8; 4 for the 64-row blocks of w4's edge tiles).
// RD = how many 16-row passes ahead the residual rows (fp32, HBM) are requested: every pass needs 4 x 16 B per lane, the
// fragment registers of the K loop are dead here, and with one pass of look-ahead the epilogue was bound by the HBM round trip.
// S2 (HIREST_EPI_BIAS_RESID2_LNSTATS): the residual stream is kept as TWO bf16 arrays, hi = bf16(x) — which is the copy the next GEMM streams
// anyway — and lo = bf16(x - hi) (16 significand bits; p.out = lo [M, ldo], p.aux0 = hi [M, N], both in / out).  The epilogue reads 2 + 2
// bytes per element and writes 2 + 2 instead of reading 4 and writing 4 + 2: a fifth less traffic on the byte-bound residual GEMMs.
template <int NI, int NM, int RD, bool S2 = false, bool DBG = false>
__device__ __forceinline__ void epilogue_lnstats(const GemmP& p, f32x4 (&acc)[8][NI], int jc, char* stg, int Mw, int Nw, int lane) {
    const int epi_dbg = DBG ? p.epi_dbg : 0, sched = DBG ? p.sched : 0;   // knock-out / A-B switches: compile-time zero in production kernels
    const int srow = lane & 15, kg = lane >> 4, sw = lane & 7;
    const int rr = lane >> 3, rc = lane & 7;
    float* outp = reinterpret_cast<float*>(p.out);
    bf16_t* xb = reinterpret_cast<bf16_t*>(p.aux0);
    float* part = reinterpret_cast<float*>(p.aux1);
    const int G = (p.N + 63) >> 6;
    f32x4 bv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = Nw + n * 16 + 4 * kg;
        bv[n] = (p.bias && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int n0 = Nw + rc * 4, n1 = Nw + 32 + rc * 4;               // this lane's columns in the two 32-column halves
    bf16_t* xlo = reinterpret_cast<bf16_t*>(p.out);                  // S2 only
    auto load_res = [&](int mi, f32x4 (&o)[2][2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = Mw + mi * 16 + it * 8 + rr;
            if constexpr (S2) {
                // 16 B per lane at the columns the stores below use (even lanes: n0 .. n0 + 7, odd lanes: n1 - 4 .. n1 + 3 — a row's 64
                // columns are one 128-B line per array), then the stores' DPP exchange backwards: an even lane keeps its first half as its
                // n0 values and takes its n1 values from the odd neighbour's first half; an odd lane keeps its second half (n1) and takes
                // its n0 values from the even neighbour's second half
                const int64_t mr = m < p.M ? m : p.M - 1;
                const int col = (rc & 1) ? n1 - 4 : n0;
                union { bf16x8 v; bf16x4 h[2]; float f[4]; } hv, lv;
                hv.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; lv.v = hv.v;
                if (col + 8 <= p.N && !(epi_dbg & 1)) {
                    if (sched & 2) {                               // A/B (hirest_gemm_debug_mode bit 17): cached instead of streaming loads
                        hv.v = *reinterpret_cast<const bf16x8*>(xb + mr * p.N + col);
                        lv.v = *reinterpret_cast<const bf16x8*>(xlo + mr * p.ldo + col);
                    } else {
                        hv.v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(xb + mr * p.N + col));
                        lv.v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(xlo + mr * p.ldo + col));
                    }
                }
                // (raw bits into the ring: decoding here would wait for the loads and undo the look-ahead — decode_s2 runs when the pass is due)
                o[0][it] = __builtin_bit_cast(f32x4, hv.v);
                o[1][it] = __builtin_bit_cast(f32x4, lv.v);
                continue;
            }
            const float* row = outp + (int64_t)(m < p.M ? m : p.M - 1) * p.ldo;
            o[0][it] = (n0 < p.N && !(epi_dbg & 1)) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + n0)) : f32x4{0.f, 0.f, 0.f, 0.f};
            o[1][it] = (n1 < p.N && !(epi_dbg & 1)) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + n1)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 ring[RD + 1][2][2];                                        // pass mi lives in ring[mi % (RD + 1)]
#pragma unroll
    for (int a = 0; a < RD && a < NM; ++a) load_res(a, ring[a]);
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        if (mi + RD < NM) load_res(mi + RD, ring[(mi + RD) % (RD + 1)]);
        f32x4 (&oc)[2][2] = ring[mi % (RD + 1)];
        if constexpr (S2) {                                          // ring slot: [0][it] = 8 hi values, [1][it] = 8 lo values of this lane's 16-B loads
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                union { bf16x8 v; bf16x4 h[2]; float f[4]; } hv, lv;
                hv.v = __builtin_bit_cast(bf16x8, oc[0][it]); lv.v = __builtin_bit_cast(bf16x8, oc[1][it]);
                union { bf16x4 v; float f[2]; } hs, hr, ls, lr;
                hs.v = (rc & 1) ? hv.h[0] : hv.h[1];
                ls.v = (rc & 1) ? lv.h[0] : lv.h[1];
                hr.f[0] = dpp_xor1(hs.f[0]); hr.f[1] = dpp_xor1(hs.f[1]);
                lr.f[0] = dpp_xor1(ls.f[0]); lr.f[1] = dpp_xor1(ls.f[1]);
                const bf16x4 h0 = (rc & 1) ? hr.v : hv.h[0], h1 = (rc & 1) ? hv.h[1] : hr.v;
                const bf16x4 l0 = (rc & 1) ? lr.v : lv.h[0], l1 = (rc & 1) ? lv.h[1] : lr.v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { oc[0][it][e] = (float)h0[e] + (float)l0[e]; oc[1][it][e] = (float)h1[e] + (float)l1[e]; }
            }
        }
        f32x4 wv[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
                *reinterpret_cast<f32x4*>(stg + srow * 128 + (((nn * 4 + kg) ^ sw) << 4)) = acc[mi][jc + h * 2 + nn] + bv[h * 2 + nn];
            HX_LDS_ORDER();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 8 + rr;
                wv[h][it] = *reinterpret_cast<const f32x4*>(stg + r * 128 + ((rc ^ (r & 7)) << 4)) + oc[h][it];
            }
            HX_LDS_ORDER();
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = Mw + mi * 16 + it * 8 + rr;
            const bool okm = m < p.M, ok0 = okm && n0 < p.N, ok1 = okm && n1 < p.N;
            if constexpr (!S2) {
                if (ok0 && !(epi_dbg & 2)) __builtin_nontemporal_store(wv[0][it], reinterpret_cast<f32x4*>(outp + (int64_t)m * p.ldo + n0));
                if (ok1 && !(epi_dbg & 2)) __builtin_nontemporal_store(wv[1][it], reinterpret_cast<f32x4*>(outp + (int64_t)m * p.ldo + n1));
            }
            union { bf16x4 v; float f[2]; } b0, b1, snd, rcv;
            float ps = 0.f, pq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b0.v[e] = (bf16_t)wv[0][it][e]; b1.v[e] = (bf16_t)wv[1][it][e];
                const float f0 = ok0 ? (float)b0.v[e] : 0.f, f1 = ok1 ? (float)b1.v[e] : 0.f;
                ps += f0 + f1; pq = fmaf(f0, f0, fmaf(f1, f1, pq));
            }
            snd.v = (rc & 1) ? b0.v : b1.v;                          // odd lanes hand over their left half, even lanes their right half
            rcv.f[0] = dpp_xor1(snd.f[0]); rcv.f[1] = dpp_xor1(snd.f[1]);
            union { bf16x8 v; float f[4]; } w8;
            int col;
            if (rc & 1) { w8.f[0] = rcv.f[0]; w8.f[1] = rcv.f[1]; w8.f[2] = b1.f[0]; w8.f[3] = b1.f[1]; col = n1 - 4; }
            else        { w8.f[0] = b0.f[0]; w8.f[1] = b0.f[1]; w8.f[2] = rcv.f[0]; w8.f[3] = rcv.f[1]; col = n0; }
            if (okm && col + 8 <= p.N && !(epi_dbg & 4)) __builtin_nontemporal_store(w8.v, reinterpret_cast<bf16x8*>(xb + (int64_t)m * p.N + col));
            if constexpr (S2) {                                      // lo = bf16(x - hi), regrouped like hi: 16 B per lane, whole lines per row
                union { bf16x4 v; float f[2]; } l0, l1, ls, lr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    l0.v[e] = (bf16_t)(wv[0][it][e] - (float)b0.v[e]);
                    l1.v[e] = (bf16_t)(wv[1][it][e] - (float)b1.v[e]);
                }
                ls.v = (rc & 1) ? l0.v : l1.v;
                lr.f[0] = dpp_xor1(ls.f[0]); lr.f[1] = dpp_xor1(ls.f[1]);
                union { bf16x8 v; float f[4]; } q8;
                if (rc & 1) { q8.f[0] = lr.f[0]; q8.f[1] = lr.f[1]; q8.f[2] = l1.f[0]; q8.f[3] = l1.f[1]; }
                else        { q8.f[0] = l0.f[0]; q8.f[1] = l0.f[1]; q8.f[2] = lr.f[0]; q8.f[3] = lr.f[1]; }
                if (okm && col + 8 <= p.N && !(epi_dbg & 2)) __builtin_nontemporal_store(q8.v, reinterpret_cast<bf16x8*>(xlo + (int64_t)m * p.ldo + col));
            }
            ps = sum8(ps); pq = sum8(pq);
            if (rc == 0 && okm && Nw < p.N && !(epi_dbg & 8)) *reinterpret_cast<f32x2*>(part + ((int64_t)m * G + (Nw >> 6)) * 2) = f32x2{ps, pq};
        }
    }
}

// (mean, rstd) of rows Mw + 2*lane and Mw + 2*lane + 1 for the LN-fold consumers (gemm_p256 brings them in by LDS-DMA at
// the start of a tile instead, so the latency hides behind the K loop)
__device__ __forceinline__ f32x4 load_row_stats(const GemmP& p, int Mw, int lane) {
    const float* st = reinterpret_cast<const float*>(p.aux0);
    const int r0 = Mw + 2 * lane;
    const f32x2 v0 = *reinterpret_cast<const f32x2*>(st + 2 * (int64_t)(r0 < p.M ? r0 : p.M - 1));
    const f32x2 v1 = *reinterpret_cast<const f32x2*>(st + 2 * (int64_t)(r0 + 1 < p.M ? r0 + 1 : p.M - 1));
    return f32x4{v0[0], v0[1], v1[0], v1[1]};
}

// SREG (gemm_d2, whose two workgroups per CU leave no LDS for them): the (mean, rstd) pairs of the lane's NM rows go
// straight from global memory (L2-resident, written by hirest_ln_stats_finalize) into registers at the start of the epilogue.
template <int EPI, int NI, bool PRE = false, int NM = 8, int RD = 1, bool SREG = false, bool DBG = false>   // PRE: the caller has already brought the row statistics into LDS
__device__ __forceinline__ void epilogue_p(const GemmP& p, f32x4 (&acc)[8][NI], char* stg, int Mw, int Nw, int lane) {
    if constexpr (EPI == HIREST_EPI_BIAS_RESID_LNSTATS_F32 || EPI == HIREST_EPI_BIAS_RESID2_LNSTATS) {
#pragma unroll
        for (int jc = 0; jc < NI; jc += 4)                            // one 64-column group at a time
            if (Nw + jc * 16 < p.N) epilogue_lnstats<NI, NM, RD, EPI == HIREST_EPI_BIAS_RESID2_LNSTATS, DBG>(p, acc, jc, stg, Mw, Nw + jc * 16, lane);
        return;
    }
    constexpr bool FOLD = epi_is_lnfold(EPI);
    constexpr bool OUT_BF16 = (EPI == HIREST_EPI_BIAS_BF16 || EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_BIAS_QGELU_BF16 || FOLD);
    const int srow = lane & 15, kg = lane >> 4, sw = lane & 7;   // sw = srow & 7
    const int rr = lane >> 3, rc = lane & 7;                     // read-back: row (it*8 + rr), 16-B chunk rc
    if constexpr (OUT_BF16) {
        bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
        f32x2 mrs[SREG ? NM : 1];
        if constexpr (FOLD && SREG) {
            const float* st = reinterpret_cast<const float*>(p.aux0);
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) {
                const int r = Mw + mi * 16 + srow;
                mrs[mi] = *reinterpret_cast<const f32x2*>(st + 2 * (int64_t)(r < p.M ? r : p.M - 1));
            }
        } else if constexpr (FOLD) {   // (mean, rstd) of this wave's 128 rows -> LDS behind the staging area (one 16-B load per lane)
            if constexpr (!PRE) *reinterpret_cast<f32x4*>(stg + P_STG + 16 * lane) = load_row_stats(p, Mw, lane);
            HX_LDS_ORDER();
        }
#pragma unroll
        for (int jp = 0; jp < NI; jp += 4) {                     // 64-column groups
            f32x4 bv[4], sv[FOLD ? 4 : 1];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = Nw + (jp + n) * 16 + 4 * kg;
                bv[n] = (p.bias && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (FOLD)
                    sv[n] = col < p.N ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.aux1) + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            f32x2 mr_next = {0.f, 1.f};
            if constexpr (FOLD && !SREG) mr_next = *reinterpret_cast<const f32x2*>(stg + P_STG + srow * 8);
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) {
                f32x2 mr = mr_next;                                    // (mean, rstd) of row mi*16 + srow, read one pass ahead
                if constexpr (FOLD && SREG) mr = mrs[mi];
                if constexpr (FOLD && !SREG) { if (mi + 1 < NM) mr_next = *reinterpret_cast<const f32x2*>(stg + P_STG + ((mi + 1) * 16 + srow) * 8); }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f32x4 v;
                    if constexpr (FOLD) {   // LayerNorm folded into the GEMM: rstd * x~ W'^T + (b' - rstd * mean * rowsum(W'))
                        const float c = -mr[0] * mr[1];                // whole-vector forms: two v_pk_fma_f32 each
                        v = acc[mi][jp + n] * f32x4{mr[1], mr[1], mr[1], mr[1]} + (sv[n] * f32x4{c, c, c, c} + bv[n]);
                    } else {
                        v = acc[mi][jp + n] + bv[n];
                    }
                    if constexpr (EPI == HIREST_EPI_BIAS_GELU_BF16 || EPI == HIREST_EPI_LNFOLD_GELU_BF16) {
                        const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
                        v = f32x4{g0[0], g0[1], g1[0], g1[1]};
                    }
                    if constexpr (EPI == HIREST_EPI_BIAS_QGELU_BF16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
                    *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((n * 2 + (kg >> 1)) ^ sw) << 4) + (kg & 1) * 8) = o;
                }
                HX_LDS_ORDER();
                const int mb = Mw + mi * 16;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = it * 8 + rr;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * 128 + ((rc ^ (r & 7)) << 4));
                    const int m = mb + r, n = Nw + jp * 16 + rc * 8;
                    if (m < p.M) {
                        bf16_t* dst = outp + (int64_t)m * p.ldo + n;
                        if (n + 8 <= p.N) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
                        else if (n + 4 <= p.N) *reinterpret_cast<bf16x4*>(dst) = bf16x4{v[0], v[1], v[2], v[3]};
                    }
                }
                HX_LDS_ORDER();
            }
        }
    } else if constexpr (EPI == HIREST_EPI_BIAS_GELU_SPLIT2) {
        // GELU (erf form, fp32) of acc + bias, stored as bf16 hi | lo in the split operand format: a 32-column group of the output is one
        // 64-element (128-B) block of the [M, 2N] row — exactly one staging row — so the read-back leaves as whole lines like the bf16 path
        bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
        for (int jp = 0; jp < NI; jp += 2) {                     // 32-column groups
            if (Nw + jp * 16 >= p.N) continue;                   // (N % 32 == 0: a group is inside or outside as a whole)
            f32x4 bv[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = Nw + (jp + n) * 16 + 4 * kg;
                bv[n] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 v = acc[mi][jp + n] + bv[n];
                    bf16x4 hi, lo;
#if HIREST_X3_GELU_POLY
                    // the bf16 towers' erf-form GELU (common.h: max abs error 1.1e-6): an fp32 erff here costs the epilogue 2 ms per launch
                    const f32x2 g0 = gelu_erf2(f32x2{v[0], v[1]}), g1 = gelu_erf2(f32x2{v[2], v[3]});
                    const f32x4 gv = {g0[0], g0[1], g1[0], g1[1]};
#else
                    f32x4 gv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)gv[e]; lo[e] = (bf16_t)(gv[e] - (float)hi[e]); }
                    *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((n * 2 + (kg >> 1)) ^ sw) << 4) + (kg & 1) * 8) = hi;
                    *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((4 + n * 2 + (kg >> 1)) ^ sw) << 4) + (kg & 1) * 8) = lo;
                }
                HX_LDS_ORDER();
                const int mb = Mw + mi * 16;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = it * 8 + rr;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * 128 + ((rc ^ (r & 7)) << 4));
                    const int m = mb + r;
                    if (m < p.M) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(outp + (int64_t)m * p.ldo + 2 * (Nw + jp * 16) + rc * 8));
                }
                HX_LDS_ORDER();
            }
        }
    } else {
        float* outp = reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int jp = 0; jp < NI; jp += 2) {                     // 32-column groups
            f32x4 bv[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = Nw + (jp + n) * 16 + 4 * kg;
                bv[n] = (p.bias && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // the 8 operand loads of each 64x32 block (residual / pos rows) are issued before the first pass: the
            // residual stream lives in HBM and a load-per-pass schedule left the epilogue latency-bound
            // (proj: 1.23 ms with the residual read vs 0.92 ms without).
            const int n = Nw + jp * 16 + rc * 4;
#pragma unroll
            for (int mh = 0; mh < NM; mh += 4) {                  // 64 rows per batch
                f32x4 o[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = Mw + mh * 16 + it * 8 + rr;
                    const bool ok = m < p.M && n < p.N;
                    if constexpr (EPI == HIREST_EPI_PATCH_POS_F32) {
                        const int mm = ok ? m : 0;
                        const int pp = mm % p.P;
                        o[it] = ok ? *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(1 + pp) * p.N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    } else if constexpr (EPI == HIREST_EPI_BIAS_RESID_F32) {
                        o[it] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(outp + (int64_t)m * p.ldo + n)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int mi = mh; mi < mh + 4; ++mi) {
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) {
                        const f32x4 v = acc[mi][jp + nn] + bv[nn];
                        *reinterpret_cast<f32x4*>(stg + srow * 128 + (((nn * 4 + kg) ^ sw) << 4)) = v;
                    }
                    HX_LDS_ORDER();
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int r = it * 8 + rr;
                        f32x4 w = *reinterpret_cast<const f32x4*>(stg + r * 128 + ((rc ^ (r & 7)) << 4));
                        if constexpr (EPI != HIREST_EPI_BIAS_F32) w += o[(mi - mh) * 2 + it];
                        const int m = Mw + mi * 16 + r;
                        if (m < p.M && n < p.N) {
                            int64_t off;
                            if constexpr (EPI == HIREST_EPI_PATCH_POS_F32) {
                                const int b = m / p.P, pp = m - b * p.P;
                                off = ((int64_t)b * (p.P + 1) + 1 + pp) * p.ldo + n;
                            } else {
                                off = (int64_t)m * p.ldo + n;
                            }
                            __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(outp + off));
                        }
                    }
                    HX_LDS_ORDER();
                }
            }
        }
    }
}

}  // namespace
