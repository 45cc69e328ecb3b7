// Flash attention of the bf16x3 vision tower (tower_x3.hip): softmax(q k^T * scale) v over fp32 q / k / v rows, both matrix products formed
// like the tower's GEMMs — from bf16 hi + lo splits of both operands, three v_mfma_f32_32x32x16_bf16 each (x_hi y_lo + x_lo y_hi + x_hi y_hi,
// fp32 accumulation: 16-bit products) — with the softmax in fp32.  Replaces the exact-fp32 attention (joint.hip: v_mfma_f32_32x32x2_f32,
// 6144 matrix-pipe cycles per 32 x 32 tile of scores) in that mode: 1152 cycles per tile, same online-softmax structure.
//   reference: EVA_clip/vit_model.py:127-147 (q k^T, softmax, @ v); no mask, no causal term (the vision tower only).
//
// One wave owns 32 queries (S^T = K Q^T puts the query in the lane: row max / sum = local reduce + one half-wave exchange); a block of NW
// waves shares the K / V tiles of 32 keys, staged global -> registers -> LDS (double-buffered, one barrier per tile) as bf16 hi / lo images:
//   K image [32 keys][DHP + 8] (A operand of S^T: 8 consecutive d per lane, one ds_read_b128), V image TRANSPOSED [DHP][32 keys + 4]
//   (A operand of O^T = V^T P^T: keys contiguous per d, two ds_read_b64 per fragment).
// P stays in registers: the C layout of S^T (lane = query; register r <-> key (r & 3) + 8 (r >> 2) + 4 half) is re-used as the B operand
// of the second product by giving k-slot (half, i) of MFMA step s2 the key 16 s2 + 4 half + (i & 3) + 8 (i >> 2) on BOTH operands.
#include "common.h"
#include <atomic>

namespace {

__device__ __forceinline__ void split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (bf16_t)v[e]; lo[e] = (bf16_t)(v[e] - (float)hi[e]); }
}

template <int DHP, int NW>
__global__ __launch_bounds__(64 * NW, NW >= 6 ? 1 : 2) void attention_x3_kernel(const float* __restrict__ qp, int64_t ldq, const float* __restrict__ kp,
                                                                 const float* __restrict__ vp, int64_t ldkv, float* __restrict__ out,
                                                                 bf16_t* __restrict__ out2, int Tq, int T, int H, int dh, float scale,
                                                                 long long* __restrict__ trace) {
    constexpr int KLD = DHP + 8, VLD = 36, NO = DHP / 32, NS = DHP / 16;     // NS: 16-deep steps of the score product
    constexpr int KSZ = 32 * KLD, VSZ = DHP * VLD;
    __shared__ __attribute__((aligned(16))) bf16_t Kh[2 * KSZ];             // two tiles: tile t + 1 is stored while tile t is multiplied
    __shared__ __attribute__((aligned(16))) bf16_t Kl[2 * KSZ];
    __shared__ __attribute__((aligned(16))) bf16_t Vh[2 * VSZ];
    __shared__ __attribute__((aligned(16))) bf16_t Vl[2 * VSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int QB = 32 * NW, NT = 64 * NW;
    const int qblocks = (Tq + QB - 1) / QB;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
    const int b = bh / H, h = bh - b * H;
    const int D = H * dh;
    const float* qbase = qp + (int64_t)b * Tq * ldq + h * dh;
    const float* kbase = kp + (int64_t)b * T * ldkv + h * dh;
    const float* vbase = vp + (int64_t)b * T * ldkv + h * dh;
    const int q = qb * QB + wave * 32 + l31;
    const bool qvalid = q < Tq;
    const bool wave_active = qb * QB + wave * 32 < Tq;               // a wave without queries only helps staging the tiles (wave-uniform)
    // Q^T fragments (B operand of S^T): lane (query, half) holds d = 16 s + 8 half .. + 7 for s = 0 .. NS - 1, as bf16 hi and lo
    bf16x8 qh[NS], ql[NS];
    {
        const float* qrow = qbase + (int64_t)(qvalid ? q : Tq - 1) * ldq;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int d = 16 * s + 8 * half + 4 * g;
                f32x4 v = *reinterpret_cast<const f32x4*>(qrow + (d + 4 <= dh ? d : dh - 4));       // dh % 4 == 0
                if (!(qvalid && d < dh)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x4 a, c;
                split4(v, a, c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { qh[s][4 * g + e] = a[e]; ql[s][4 * g + e] = c[e]; }
            }
        }
    }
    f32x16 o[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
    float mrun = -3.0e38f, lrun = 0.f;
    // K / V tile staging through registers, one tile ahead of the LDS images, which are one tile ahead of the MFMAs.
    //   K: thread -> (key = i / (DHP / 4), 4 consecutive d): whole 128-B lines per wave-load, 8-B hi / lo stores along a row: conflict-free.
    //   V: its image is TRANSPOSED ([d][key], 2-byte stores), so a wave-load covers 16 keys x 4 float4 instead — lane = key (16) + 16 chunk —
    //      which puts a store instruction's 64 lanes on 32 distinct banks (key / 2 + 8 chunk words); with K's mapping 24 lanes of one key
    //      hit 4 banks (57 % of the LDS cycles were conflicts).
    constexpr int NV4 = 32 * (DHP / 4), NLD = (NV4 + NT - 1) / NT;
    constexpr int VINS = 2 * (DHP / 16);                              // V wave-instructions per tile: (16-key half) x (group of 4 chunks)
    constexpr int NLV = (VINS + NW - 1) / NW;
    f32x4 kreg[NLD], vreg[NLV];
    auto v_slot = [&](int u, int& kr, int& c) {                       // (wave-uniform instruction index; false: nothing to do)
        const int t = wave + NW * u;
        kr = 16 * (t & 1) + (lane & 15);
        c = 16 * (t >> 1) + 4 * (lane >> 4);
        return t < VINS;
    };
    auto fetch_kv = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            int i = tid + NT * u; i = i < NV4 ? i : NV4 - 1;
            const int kr = i / (DHP / 4), c = (i - kr * (DHP / 4)) * 4;
            const int key = k0 + kr < T ? k0 + kr : T - 1;
            kreg[u] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * ldkv + (c < dh ? c : dh - 4));
        }
#pragma unroll
        for (int u = 0; u < NLV; ++u) {
            int kr, c;
            v_slot(u, kr, c);
            const int key = k0 + kr < T ? k0 + kr : T - 1;
            c = c < DHP ? c : DHP - 4;
            vreg[u] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * ldkv + (c < dh ? c : dh - 4));
        }
    };
    auto store_kv = [&](int k0, int buf) {                            // registers (tile at k0) -> LDS images `buf`; rows past T and padded columns: zeros
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + NT * u;
            if (i < NV4) {
                const int kr = i / (DHP / 4), c = (i - kr * (DHP / 4)) * 4;
                bf16x4 a, l;
                split4((c < dh && k0 + kr < T) ? kreg[u] : z, a, l);
                *reinterpret_cast<bf16x4*>(&Kh[buf * KSZ + kr * KLD + c]) = a;
                *reinterpret_cast<bf16x4*>(&Kl[buf * KSZ + kr * KLD + c]) = l;
            }
        }
#pragma unroll
        for (int u = 0; u < NLV; ++u) {
            int kr, c;
            if (v_slot(u, kr, c)) {
                bf16x4 a, l;
                split4((c < dh && k0 + kr < T) ? vreg[u] : z, a, l);
#pragma unroll
                for (int e = 0; e < 4; ++e) { Vh[buf * VSZ + (c + e) * VLD + kr] = a[e]; Vl[buf * VSZ + (c + e) * VLD + kr] = l[e]; }
            }
        }
    };
    fetch_kv(0);
    store_kv(0, 0);
    if (32 < T) fetch_kv(32);
    int buf = 0;
    // timing tool (tools/attn_x3_trace.py): workgroup 0 stamps the shader clock at the phase boundaries of every tile: [tile][wave][8]
    auto stamp = [&](int k0, int slot) {
        if (trace && blockIdx.x == 0) {
            const long long t = __builtin_readcyclecounter();
            if (lane == 0) trace[((k0 >> 5) * 16 + wave) * 8 + slot] = t;
        }
    };
    for (int k0 = 0; k0 < T; k0 += 32, buf ^= 1) {
        stamp(k0, 0);
        // one barrier per tile: it publishes this tile's images (stored during the previous iteration) and tells every wave that the
        // other buffer — read during the previous iteration — is free for the next tile
        __syncthreads();
        stamp(k0, 1);
        if (trace && blockIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(k0, 6); }   // (tool only: the tile's loads have landed)
        if (k0 + 32 < T) store_kv(k0 + 32, buf ^ 1);
        stamp(k0, 7);                                                 // (split + LDS stores done: the stamp drains lgkmcnt)
        if (k0 + 64 < T) fetch_kv(k0 + 64);
        stamp(k0, 2);
        const bf16_t* Khb = Kh + buf * KSZ; const bf16_t* Klb = Kl + buf * KSZ;
        const bf16_t* Vhb = Vh + buf * VSZ; const bf16_t* Vlb = Vl + buf * VSZ;
        if (!wave_active) continue;
        // ---- S^T = K Q^T: A = K[key = l31][d = 16 s + 8 half ..], B = Q^T; small terms first
        f32x16 st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&Khb[l31 * KLD + 16 * s + 8 * half]);
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(&Klb[l31 * KLD + 16 * s + 8 * half]);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], st, 0, 0, 0);
        }
        stamp(k0, 3);
        // ---- online softmax (fp32): st[r] = score of key k0 + (r & 3) + 8 (r >> 2) + 4 half for query l31
        float tmax = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sv = key < T ? st[r] * scale : -3.0e38f;
            st[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        { float pa = tmax, pb = tmax; lane_swap32(pa, pb); tmax = fmaxf(pa, pb); }
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
        bf16x8 ph[2], pl[2];                                        // P^T fragments of the two 16-key MFMA steps: registers 8 s2 .. 8 s2 + 7
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pz = __expf(st[r] - mnew);
            psum += pz;
            const bf16_t hi = (bf16_t)pz;
            ph[r >> 3][r & 7] = hi;
            pl[r >> 3][r & 7] = (bf16_t)(pz - (float)hi);
        }
        { float pa = psum, pb = psum; lane_swap32(pa, pb); psum = pa + pb; }
        lrun = __builtin_fmaf(lrun, alpha, psum);
        mrun = mnew;
#pragma unroll
        for (int j = 0; j < NO; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
        stamp(k0, 4);
        // ---- O^T += V^T P^T: A = V^T[d = l31 + 32 j][key slots of this half], k-slot i <-> key 16 s2 + 4 half + (i & 3) + 8 (i >> 2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const int ro = (32 * j + l31) * VLD + 16 * s2 + 4 * half;
                union { bf16x8 v; bf16x4 h[2]; } vh, vl;
                vh.h[0] = *reinterpret_cast<const bf16x4*>(&Vhb[ro]); vh.h[1] = *reinterpret_cast<const bf16x4*>(&Vhb[ro + 8]);
                vl.h[0] = *reinterpret_cast<const bf16x4*>(&Vlb[ro]); vl.h[1] = *reinterpret_cast<const bf16x4*>(&Vlb[ro + 8]);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh.v, pl[s2], o[j], 0, 0, 0);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl.v, ph[s2], o[j], 0, 0, 0);
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh.v, ph[s2], o[j], 0, 0, 0);
            }
        }
        stamp(k0, 5);
    }
    if (!qvalid) return;
    const float inv = 1.0f / lrun;
    float* orow = out ? out + ((int64_t)b * Tq + q) * D + h * dh : nullptr;
    bf16_t* orow2 = out2 ? out2 + ((int64_t)b * Tq + q) * 2 * D : nullptr;      // split operand format of the next GEMM (hirest_split2_bf16)
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // O^T rows (reg & 3) + 8 (reg >> 2) + 4 half = d
            const int d = 32 * j + 8 * g + 4 * half;
            if (d < dh) {
                const f32x4 y = {o[j][4 * g] * inv, o[j][4 * g + 1] * inv, o[j][4 * g + 2] * inv, o[j][4 * g + 3] * inv};
                if (orow) *reinterpret_cast<f32x4*>(orow + d) = y;
                if (orow2) {            // (four consecutive columns from a multiple of 4 never straddle a 32-column block)
                    const int col = h * dh + d;
                    bf16x4 hi, lo;
                    split4(y, hi, lo);
                    bf16_t* o2 = orow2 + (col >> 5) * 64 + (col & 31);
                    *reinterpret_cast<bf16x4*>(o2) = hi;
                    *reinterpret_cast<bf16x4*>(o2 + 32) = lo;
                }
            }
        }
}

template <int DHP, class... Args>
void launch_x3(int nw, dim3 grid, hipStream_t s, Args... args) {
    if (nw == 3) hipLaunchKernelGGL((attention_x3_kernel<DHP, 3>), grid, dim3(192), 0, s, args...);
    else if (nw == 8) hipLaunchKernelGGL((attention_x3_kernel<DHP, 8>), grid, dim3(512), 0, s, args...);
    else if (nw == 9) hipLaunchKernelGGL((attention_x3_kernel<DHP, 9>), grid, dim3(576), 0, s, args...);
    else hipLaunchKernelGGL((attention_x3_kernel<DHP, 4>), grid, dim3(256), 0, s, args...);
}
long long* g_x3_trace = nullptr;   // device buffer for the phase stamps (hirest_attention_x3_debug_trace; timing tool only)
std::atomic<int> g_x3_waves{0};      // 0 automatic; 3 / 4 / 8 / 9 force (hirest_attention_x3_select_waves)

}  // namespace

static int attention_x3(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, bf16_t* out2, int32_t B,
                        int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, void* stream) {
    if (!q || !k || !v || (!out && !out2) || B <= 0 || Tq <= 0 || Tk <= 0 || H <= 0) return HIREST_E_BADARG;
    if (out2 && ((H * dh) % 32 != 0 || (reinterpret_cast<uintptr_t>(out2) & 7))) return HIREST_E_SHAPE;
    if (dh <= 0 || dh % 4 != 0 || dh > 96 || ldq % 4 != 0 || ldkv % 4 != 0) return HIREST_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15)   // (out may be NULL)
        return HIREST_E_SHAPE;
    const int waves = (Tq + 31) / 32;
    int nw = (waves + 2) / 3 * 3 < (waves + 3) / 4 * 4 ? 3 : 4;               // three-wave blocks when they waste fewer waves (257 queries: 9)
    // exactly nine waves of queries (the ViT's 257 tokens): one workgroup per (frame, head) stages every K / V tile once instead of three
    // times (5.5 -> 2.2 GB read per 512-frame launch; 3.21 -> 3.07 ms at 1024 frames: the kernel is bound by instruction issue, not by bytes)
    if (waves == 9 && B * H >= 512) nw = 9;
    if (g_x3_waves) nw = g_x3_waves;
    const dim3 grid((unsigned)((int64_t)B * H * ((waves + nw - 1) / nw)));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dh <= 32) launch_x3<32>(nw, grid, s, q, ldq, k, v, ldkv, out, out2, (int)Tq, (int)Tk, (int)H, (int)dh, scale, g_x3_trace);
    else if (dh <= 64) launch_x3<64>(nw, grid, s, q, ldq, k, v, ldkv, out, out2, (int)Tq, (int)Tk, (int)H, (int)dh, scale, g_x3_trace);
    else launch_x3<96>(nw, grid, s, q, ldq, k, v, ldkv, out, out2, (int)Tq, (int)Tk, (int)H, (int)dh, scale, g_x3_trace);
    return hirest_launch_status();
}

extern "C" int hirest_attention_x3_debug_trace(int64_t* device_buffer) {   // [tiles <= 16][waves <= 16][8] int64 on the device, or NULL (off)
    g_x3_trace = reinterpret_cast<long long*>(device_buffer);
    return 0;
}

extern "C" int hirest_attention_x3_select_waves(int32_t waves) {
    if (waves != 0 && waves != 3 && waves != 4 && waves != 8 && waves != 9) return HIREST_E_BADARG;
    g_x3_waves = waves;
    return 0;
}

extern "C" int hirest_attention_x3_qkv(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, int32_t B,
                                       int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, void* stream) {
    return attention_x3(q, ldq, k, v, ldkv, out, nullptr, B, Tq, Tk, H, dh, scale, stream);
}
// the same with the output written as the split operand [B * Tq, 2 * H * dh] bf16 of the GEMM that follows (hirest_split2_bf16's format)
extern "C" int hirest_attention_x3_qkv_split2(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, hirest_bf16* out2,
                                              int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, void* stream) {
    return attention_x3(q, ldq, k, v, ldkv, nullptr, reinterpret_cast<bf16_t*>(out2), B, Tq, Tk, H, dh, scale, stream);
}
