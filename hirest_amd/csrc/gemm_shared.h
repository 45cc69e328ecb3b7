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
    int flat = 0;        // gemm_t128x3: flat tile list cut into 8 per-XCD chunks instead of the row-panel split (few row panels)
    int ksplit = 1;      // gemm_t128x3: K slices per tile (flat mapping only); > 1: raw partial tiles to `part`, splitk_reduce_kernel finishes
    float* part = nullptr;
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
#ifndef HIREST_S2_BUFFER_EPILOGUE
#define HIREST_S2_BUFFER_EPILOGUE 1      // 0: round 5's flat-address form of the two-array residual epilogue (A/B builds)
#endif
#ifndef HIREST_S2_RD
#define HIREST_S2_RD 2                   // residual look-ahead (16-row passes) of the two-array residual epilogue in gemm_pq256
#endif
constexpr int P_STG = 2048;    // epilogue staging bytes per wave: 16 rows x 128 B
constexpr bool epi_is_lnfold(int epi) { return epi == HIREST_EPI_LNFOLD_BF16 || epi == HIREST_EPI_LNFOLD_GELU_BF16; }
// the LN-fold consumers keep the (mean, rstd) pairs of the wave's 128 rows behind their staging area (1 KiB) and, in the ping-pong kernels, the
// tile's bias | column-sum slices behind those (P_BCS: 64 + 64 floats, brought in by one LDS-DMA piece at the start of the tile)
constexpr int P_BCS = P_STG + 1024;
// (the straight-line two-array residual epilogue stages both 32-column halves of a pass at once: 2 x P_STG, which brings the kernel to 160 KiB)
constexpr int p_stg_bytes(int epi) {
    return epi_is_lnfold(epi) ? P_STG + 2048 : (epi == HIREST_EPI_BIAS_RESID2_LNSTATS && HIREST_S2_BUFFER_EPILOGUE) ? 2 * P_STG : P_STG;
}

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
// Buffer (range-checked) accesses for the two-array residual epilogue.  Why: with flat global loads / stores every ragged-edge predicate is a
// branch (s_cbranch_execz around the access), the compiler cannot count vector-memory operations across branches, and it fell back to
// s_waitcnt vmcnt(0) — the look-ahead ring was drained three times per tile, stores included (round-5 ISA: 3 full HBM round trips per
// 128 x 64 block).  A raw buffer access with an out-of-range offset is dropped (store) or returns zeros (load) in hardware: the epilogue becomes
// straight-line code and every wait is a counted vmcnt(n) that leaves the newer loads and the stores in flight.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
constexpr unsigned HX_OOB = 0x80000000u;                            // beyond any num_records used here
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hx_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
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


// HIREST_EPI_BIAS_RESID2_LNSTATS, straight-line form (round 6): the arithmetic of epilogue_lnstats<..., S2 = true> above, operation for operation,
// with every global access a range-checked buffer access (see hx_rsrc): no predicate branches, so the compiler's waits are counted and the
// residual rows of pass mi + RD really are in flight while pass mi is processed.  Per wave and 64-column group: 8 passes of 16 rows; a pass loads
// 2 x (hi, lo) x 16 B per lane and stores 2 x (hi', lo') x 16 B + the row partials.
// Registers are what limits the look-ahead (128 accumulators + the K loop's stream state stay live), so against the flat form: both 32-column
// halves of a pass go through the staging area at once (4 KiB per wave: the kernel now uses all 160 KiB of the CU) and the 8-row sets of a pass
// are finished one after the other; the bias is added after the transpose (8 registers instead of 16; (acc + bias) + residual in the same order).
// Addressing: one descriptor per array for the tile and ONE per-lane offset per array; rows at or past M are cut by selecting an out-of-range
// per-lane offset.  The 8-row step k is a scalar offset for the LOADS only.  The 16-byte STORES carry it in the per-lane offset with a literal
// scalar offset of 0: a buffer_store_dwordx4 with an SGPR scalar offset followed directly by a VALU write of its data registers stored the NEW
// contents of the second data dword on gfx950 (measured round 6: rows of every step k >= 1 wrong in exactly the overwritten register, k = 0 —
// literal offset, where the compiler pads the documented store-data hazard — right); the compiler's hazard model exempts the SGPR-offset form.
template <int NI, int NM, int RD, bool DBG = false>
__device__ __forceinline__ void epilogue_lnstats2(const GemmP& p, f32x4 (&acc)[8][NI], int jc, char* stg, int Mw, int Nw, int lane) {
    const int epi_dbg = DBG ? p.epi_dbg : 0, sched = DBG ? p.sched : 0;
    const int srow = lane & 15, kg = lane >> 4, sw = lane & 7;
    const int rr = lane >> 3, rc = lane & 7;
    const int G = (p.N + 63) >> 6;
    const int N = p.N, ldo = (int)p.ldo;
    int rows = p.M - Mw; rows = rows < NM * 16 ? rows : NM * 16;      // > 0: the caller checked that this wave's block holds rows
    const __amdgpu_buffer_rsrc_t rs_hi = hx_rsrc(reinterpret_cast<const bf16_t*>(p.aux0) + (int64_t)Mw * N, (unsigned)rows * (unsigned)N * 2u);
    const __amdgpu_buffer_rsrc_t rs_lo = hx_rsrc(reinterpret_cast<const bf16_t*>(p.out) + (int64_t)Mw * ldo, ((unsigned)(rows - 1) * (unsigned)ldo + (unsigned)N) * 2u);
    const __amdgpu_buffer_rsrc_t rs_pt = hx_rsrc(reinterpret_cast<const float*>(p.aux1) + ((int64_t)Mw * G + (Nw >> 6)) * 2, ((unsigned)(rows - 1) * (unsigned)G + 1u) * 8u);
    const __amdgpu_buffer_rsrc_t rs_b = hx_rsrc(p.bias, p.bias ? (unsigned)N * 4u : 0u);
    const int n0 = Nw + rc * 4, n1 = Nw + 32 + rc * 4;               // this lane's columns in the two 32-column halves (after the transpose)
    // columns past N (or no bias at all): out of range -> zeros
    const f32x4 bias0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (unsigned)n0 * 4u, 0, 0));
    const f32x4 bias1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (unsigned)n1 * 4u, 0, 0));
    // 16 B per lane at the columns the stores use (even lanes: n0 .. n0 + 7, odd lanes: n1 - 4 .. n1 + 3 — a row's 64 columns are one 128-B line
    // per array); loads and stores of a pass share their offsets
    const int col = (rc & 1) ? n1 - 4 : n0;
    const bool col_ok = col + 8 <= N;
    const unsigned vh0 = col_ok ? (unsigned)(rr * N + col) * 2u : HX_OOB, vl0 = col_ok ? (unsigned)(rr * ldo + col) * 2u : HX_OOB;
    const unsigned vp0 = (rc == 0 && Nw < N) ? (unsigned)(rr * G) * 8u : HX_OOB;
    const unsigned sh = 16u * (unsigned)N, sl = 16u * (unsigned)ldo, sp = 64u * (unsigned)G;   // bytes per 8 rows
    const unsigned vh_ld = (epi_dbg & 1) ? HX_OOB : vh0, vl_ld = (epi_dbg & 1) ? HX_OOB : vl0;
    const unsigned vh_st = (epi_dbg & 4) ? HX_OOB : vh0, vl_st = (epi_dbg & 2) ? HX_OOB : vl0, vp_st = (epi_dbg & 8) ? HX_OOB : vp0;
    auto load_res = [&](int mi, u32x4 (&o)[2][2]) {                  // raw bits: decoded when the pass is due, so nothing waits here
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = mi * 2 + it;
            const bool okm = k * 8 + rr < rows;
            const unsigned vh = okm ? vh_ld : HX_OOB, vl = okm ? vl_ld : HX_OOB;
            if (DBG && (sched & 2)) {                                // A/B (hirest_gemm_debug_mode bit 17): cached instead of streaming loads
                o[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_hi, vh, (unsigned)k * sh, 0);
                o[1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, vl, (unsigned)k * sl, 0);
            } else {
                o[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_hi, vh, (unsigned)k * sh, 2);
                o[1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_lo, vl, (unsigned)k * sl, 2);
            }
        }
    };
    u32x4 ring[RD + 1][2][2];                                        // pass mi lives in ring[mi % (RD + 1)]
#pragma unroll
    for (int a = 0; a < RD && a < NM; ++a) load_res(a, ring[a]);
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        if (mi + RD < NM) load_res(mi + RD, ring[(mi + RD) % (RD + 1)]);
        __builtin_amdgcn_sched_barrier(0);                           // (free to move, the scheduler issues every pass's loads at once and spills)
        // ---- both 32-column halves of the pass -> staging (block h = the flat form's 16 x 128-B image of half h)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
                *reinterpret_cast<f32x4*>(stg + h * P_STG + srow * 128 + (((nn * 4 + kg) ^ sw) << 4)) = acc[mi][jc + h * 2 + nn];
        HX_LDS_ORDER();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = mi * 2 + it;
            const int r = it * 8 + rr;
            // residual values at (n0 .. n0 + 3) and (n1 .. n1 + 3) of row r: the stores' DPP exchange run backwards
            f32x4 oc0, oc1;
            {
                union { bf16x8 v; bf16x4 h[2]; float f[4]; } hv, lv;
                hv.v = __builtin_bit_cast(bf16x8, ring[mi % (RD + 1)][0][it]); lv.v = __builtin_bit_cast(bf16x8, ring[mi % (RD + 1)][1][it]);
                union { bf16x4 v; float f[2]; } hs, hr, ls, lr;
                hs.v = (rc & 1) ? hv.h[0] : hv.h[1];
                ls.v = (rc & 1) ? lv.h[0] : lv.h[1];
                hr.f[0] = dpp_xor1(hs.f[0]); hr.f[1] = dpp_xor1(hs.f[1]);
                lr.f[0] = dpp_xor1(ls.f[0]); lr.f[1] = dpp_xor1(ls.f[1]);
                const bf16x4 h0 = (rc & 1) ? hr.v : hv.h[0], h1 = (rc & 1) ? hv.h[1] : hr.v;
                const bf16x4 l0 = (rc & 1) ? lr.v : lv.h[0], l1 = (rc & 1) ? lv.h[1] : lr.v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { oc0[e] = (float)h0[e] + (float)l0[e]; oc1[e] = (float)h1[e] + (float)l1[e]; }
            }
            const f32x4 wv0 = (*reinterpret_cast<const f32x4*>(stg + r * 128 + ((rc ^ (r & 7)) << 4)) + bias0) + oc0;
            const f32x4 wv1 = (*reinterpret_cast<const f32x4*>(stg + P_STG + r * 128 + ((rc ^ (r & 7)) << 4)) + bias1) + oc1;
            const bool okm = k * 8 + rr < rows, ok0 = okm && n0 < N, ok1 = okm && n1 < N;
            union { bf16x4 v; float f[2]; } b0, b1, snd, rcv;
            float ps = 0.f, pq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b0.v[e] = (bf16_t)wv0[e]; b1.v[e] = (bf16_t)wv1[e];
                const float f0 = ok0 ? (float)b0.v[e] : 0.f, f1 = ok1 ? (float)b1.v[e] : 0.f;
                ps += f0 + f1; pq = fmaf(f0, f0, fmaf(f1, f1, pq));
            }
            snd.v = (rc & 1) ? b0.v : b1.v;                          // odd lanes hand over their left half, even lanes their right half
            rcv.f[0] = dpp_xor1(snd.f[0]); rcv.f[1] = dpp_xor1(snd.f[1]);
            union { u32x4 u; float f[4]; } w8;
            if (rc & 1) { w8.f[0] = rcv.f[0]; w8.f[1] = rcv.f[1]; w8.f[2] = b1.f[0]; w8.f[3] = b1.f[1]; }
            else        { w8.f[0] = b0.f[0]; w8.f[1] = b0.f[1]; w8.f[2] = rcv.f[0]; w8.f[3] = rcv.f[1]; }
            __builtin_amdgcn_raw_buffer_store_b128(w8.u, rs_hi, okm ? vh_st + (unsigned)k * sh : HX_OOB, 0, 2);
            union { bf16x4 v; float f[2]; } l0, l1, ls, lr;          // lo = bf16(x - hi), regrouped like hi: 16 B per lane, whole lines per row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                l0.v[e] = (bf16_t)(wv0[e] - (float)b0.v[e]);
                l1.v[e] = (bf16_t)(wv1[e] - (float)b1.v[e]);
            }
            ls.v = (rc & 1) ? l0.v : l1.v;
            lr.f[0] = dpp_xor1(ls.f[0]); lr.f[1] = dpp_xor1(ls.f[1]);
            union { u32x4 u; float f[4]; } q8;
            if (rc & 1) { q8.f[0] = lr.f[0]; q8.f[1] = lr.f[1]; q8.f[2] = l1.f[0]; q8.f[3] = l1.f[1]; }
            else        { q8.f[0] = l0.f[0]; q8.f[1] = l0.f[1]; q8.f[2] = lr.f[0]; q8.f[3] = lr.f[1]; }
            __builtin_amdgcn_raw_buffer_store_b128(q8.u, rs_lo, okm ? vl_st + (unsigned)k * sl : HX_OOB, 0, 2);
            ps = sum8(ps); pq = sum8(pq);
            union { u32x2 u; float f[2]; } st;
            st.f[0] = ps; st.f[1] = pq;
            __builtin_amdgcn_raw_buffer_store_b64(st.u, rs_pt, okm ? vp_st + (unsigned)k * sp : HX_OOB, 0, 0);
        }
        HX_LDS_ORDER();
        __builtin_amdgcn_sched_barrier(0);
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
// BLDS (with PRE, LN-fold epilogues of 64-column wave tiles): the caller has also brought bias[Nw .. Nw + 63] | colsum[Nw .. Nw + 63] into LDS at
// stg + P_BCS (columns past N clamped: their results are never stored), so the epilogue starts without a single global load.
template <int EPI, int NI, bool PRE = false, int NM = 8, int RD = 1, bool SREG = false, bool DBG = false, bool BLDS = false>   // PRE: the caller has already brought the row statistics into LDS
__device__ __forceinline__ void epilogue_p(const GemmP& p, f32x4 (&acc)[8][NI], char* stg, int Mw, int Nw, int lane) {
    if constexpr (EPI == HIREST_EPI_BIAS_RESID_LNSTATS_F32 || EPI == HIREST_EPI_BIAS_RESID2_LNSTATS) {
#pragma unroll
        for (int jc = 0; jc < NI; jc += 4)                            // one 64-column group at a time
            if (Nw + jc * 16 < p.N) {
                if constexpr (EPI == HIREST_EPI_BIAS_RESID2_LNSTATS && HIREST_S2_BUFFER_EPILOGUE) epilogue_lnstats2<NI, NM, RD, DBG>(p, acc, jc, stg, Mw, Nw + jc * 16, lane);
                else epilogue_lnstats<NI, NM, RD, EPI == HIREST_EPI_BIAS_RESID2_LNSTATS, DBG>(p, acc, jc, stg, Mw, Nw + jc * 16, lane);
            }
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
                if constexpr (FOLD && BLDS) {
                    static_assert(!BLDS || NI == 4, "one 64-column group per wave");
                    const f32x4 b = *reinterpret_cast<const f32x4*>(stg + P_BCS + (n * 16 + 4 * kg) * 4);
                    bv[n] = p.bias ? b : f32x4{0.f, 0.f, 0.f, 0.f};
                    sv[n] = *reinterpret_cast<const f32x4*>(stg + P_BCS + 256 + (n * 16 + 4 * kg) * 4);
                } else {
                    bv[n] = (p.bias && col < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (FOLD)
                        sv[n] = col < p.N ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.aux1) + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
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
