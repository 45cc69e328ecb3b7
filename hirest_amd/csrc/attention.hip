// Multi-head self-attention core for short sequences (N <= 272: the ViT's 257 tokens, CLIP's 77).
// Replaces q@k^T -> softmax -> @v (vit_model.py:131-147; nn.MultiheadAttention core with the
// additive causal mask of eva_model.py:224-230).
//
// One workgroup (8 waves) per (frame, head).  K [N x dh] and V^T [dh x N] of that head live in LDS
// for the whole block; each wave walks 16-query tiles:
//   S^T = K . Q^T      v_mfma_f32_16x16x32_bf16(a = K rows from LDS, b = Q rows straight from HBM)
//   softmax over keys  fully in registers: the C layout of S^T puts query (lane&15) in the lane and
//                      keys in (lane>>4, reg), so row max / sum are a local reduce + 2 shuffles
//   O^T = V^T . P^T    the same registers, converted to bf16, ARE the B operand (k-slots permuted
//                      consistently on the V^T side: slot j<4 -> key 32s+4g+j, j>=4 -> 32s+16+4g+j-4)
// so P never leaves registers and O^T leaves each lane with 4 consecutive head-dim values (8-B stores).
// Head dim 88 is zero-padded to 96 for the QK^T contraction; padded keys are masked to -inf.
#include "common.h"
#include <atomic>
#include "profile.h"

namespace {

template <int DH, int DP, int NT>
struct AttnCfg {
    static constexpr int NPAD = 16 * NT;
    static constexpr int KS = (NT + 1) / 2;        // 32-key steps for P.V
    static constexpr int KP = 32 * KS;
    static constexpr int KRS = DP * 2 + 16;        // K row stride (bytes), padded
    static constexpr int VRS = KP * 2 + 16;        // V^T row stride (bytes), padded
    static constexpr int LDS_BYTES = NPAD * KRS + DP * VRS;
};

template <int DH, int DP, int NT>
__global__ __launch_bounds__(512) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       int N, int H, float scale_log2e, int causal) {
    using C = AttnCfg<DH, DP, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + C::NPAD * C::KRS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * DH;

    // ---- stage K (row-major, zero padded) ----
    constexpr int KCH = DP / 8;
    for (int idx = tid; idx < C::NPAD * KCH; idx += 512) {
        const int key = idx / KCH, ch = idx - key * KCH;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < N && ch * 8 < DH) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + D + ch * 8);
        *reinterpret_cast<bf16x8*>(Ks + key * C::KRS + ch * 16) = v;
    }
    // ---- stage V transposed: Vt[d][key] ----
    constexpr int VCH = DH / 8;
    for (int idx = tid; idx < C::KP * VCH; idx += 512) {
        const int ch = idx / C::KP, key = idx - ch * C::KP;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < N) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + 2 * D + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(Vt + (ch * 8 + e) * C::VRS + key * 2) = v[e];
    }
    __syncthreads();

    const int g = lane >> 4, c16 = lane & 15;
    const int nqt = (N + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += 8) {
        const int q = qt * 16 + c16;
        const bool qvalid = q < N;
        bf16x8 qf[DP / 32];
#pragma unroll
        for (int kk = 0; kk < DP / 32; ++kk) {
            const int d = (kk * 4 + g) * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qvalid && d < DH) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * ld + d);
            qf[kk] = v;
        }
        // ---- S^T tiles ----
        f32x4 st[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DP / 32; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c16) * C::KRS + (kk * 4 + g) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], acc, 0, 0, 0);
            }
            st[t] = acc;
            if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // bound how many K fragments are hoisted (VGPR budget: 2 waves/SIMD)
        }
        // ---- mask + softmax (fp32) ----
        const int klimit = causal ? (q < N - 1 ? q : N - 1) : N - 1;   // last visible key
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = t * 16 + 4 * g + i;
                const float s = key <= klimit ? st[t][i] : -3.0e38f;
                st[t][i] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pz = exp2f((st[t][i] - mx) * scale_log2e);
                st[t][i] = pz;
                sum += pz;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // ---- P^T as bf16 B fragments ----
        bf16x8 pf[C::KS];
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf[s][i] = (bf16_t)st[2 * s][i];
                if (2 * s + 1 < NT) pf[s][4 + i] = (bf16_t)st[2 * s + 1][i];
                else pf[s][4 + i] = (bf16_t)0.f;
            }
        }
        // ---- O^T = V^T . P^T ----
#pragma unroll 1
        for (int dt = 0; dt < DP / 16; ++dt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const char* vrow = Vt + (dt * 16 + c16) * C::VRS + 8 * g;
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + s * 64);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + s * 64 + 32);
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s], acc, 0, 0, 0);
            }
            const int d = dt * 16 + 4 * g;
            if (qvalid && d < DH) {
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(acc[i] * inv);
                *reinterpret_cast<bf16x4*>(out + ((int64_t)b * N + q) * D + h * DH + d) = o;
            }
        }
    }
}

template <int DH, int DP, int NT>
int launch(const bf16_t* qkv, bf16_t* out, int B, int N, int H, float scale, int causal, hipStream_t s) {
    using C = AttnCfg<DH, DP, NT>;
    static HirestDevCfg cfg;
    auto kern = attention_kernel<DH, DP, NT>;
    if (int e = hirest_configure(kern, C::LDS_BYTES, cfg)) return e;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(512), C::LDS_BYTES, s, qkv, out, N, H, scale * 1.44269504088896340736f, causal);
    return hirest_launch_status();
}


// =================================================================================================
// v2: K and V both staged ROW-MAJOR by LDS-DMA (global_load_lds_dwordx4; no VGPR round trip, no
// scattered 2-byte transposing stores), V consumed through ds_read_b64_tr_b16 (the hardware 4x16
// transpose read): each 16-lane group reads one [4 keys][16 head-dims] block and lane i gets column i.
// LDS image: [NPAD K rows][KP V rows], DP bf16 per row (192 B for head dim 88: chunk 11 of a row is
// a duplicate of chunk 10 — it only ever meets the zeroed tail of the Q fragment / feeds discarded
// output rows).  Rows past N are clamped duplicates of row N-1 (their scores are masked to -inf).
// =================================================================================================
template <int DH, int DP, int NT>
struct AttnCfg2 {
    static constexpr int NPAD = 16 * NT;
    static constexpr int KS = (NT + 1) / 2;
    static constexpr int KP = 32 * KS;
    static constexpr int RS = DP * 2;              // row stride, bytes
    static constexpr int CPR = DP / 8;             // 16-B chunks per row
    static constexpr int NK_INSTR = NPAD * CPR / 64;
    static constexpr int NV_INSTR = KP * CPR / 64;
    static constexpr int LDS_BYTES = (NPAD + KP) * RS;
};

__device__ __forceinline__ void wait_vm_le(int n) {   // s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0,8]
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

template <int DH, int DP, int NT>
__global__ __launch_bounds__(512) void attention_kernel_v2(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          int N, int H, float scale_log2e, int causal) {
    using C = AttnCfg2<DH, DP, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + C::NPAD * C::RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * DH;

    // ---- LDS-DMA: K pieces first, then V pieces (1 KiB = 64 consecutive 16-B chunks per wave-instruction)
    int nv_mine = 0;
    for (int i = wave; i < C::NK_INSTR; i += 8) {
        const int ci = i * 64 + lane;
        int row = ci / C::CPR, c = ci - row * C::CPR;
        // K image is XOR-swizzled inside each 4-chunk group: position p holds chunk (p&~3)|((p&3)^f(row)),
        // f(row) = (-(row>>2))&3 — makes the S^T fragment reads (16 rows x one chunk per 16-lane group)
        // hit 16 distinct 16-B slots at the 192-B row stride (was a 2-way conflict).
        c = (c & ~3) | ((c & 3) ^ ((-(row >> 2)) & 3));
        row = row < N ? row : N - 1;
        c = c * 8 < DH ? c : DH / 8 - 1;
        glds16(base + (int64_t)row * ld + D + c * 8, Ks + i * 1024);
    }
    for (int i = wave; i < C::NV_INSTR; i += 8) {
        const int ci = i * 64 + lane;
        int row = ci / C::CPR, c = ci - row * C::CPR;
        row = row < N ? row : N - 1;
        c = c * 8 < DH ? c : DH / 8 - 1;
        glds16(base + (int64_t)row * ld + 2 * D + c * 8, Vs + i * 1024);
        ++nv_mine;
    }
    wait_vm_le(nv_mine);            // this wave's K pieces have landed (V may still be in flight)
    __builtin_amdgcn_s_barrier();   // ... and everybody else's

    const int g = lane >> 4, c16 = lane & 15;
    const int gk = g ^ ((-(c16 >> 2)) & 3);   // swizzled chunk-in-group for this lane's K rows
    const int nqt = (N + 15) >> 4;
    bool v_ready = false;
    for (int qt = wave; qt < nqt; qt += 8) {
        const int q = qt * 16 + c16;
        const bool qvalid = q < N;
        bf16x8 qf[DP / 32];
#pragma unroll
        for (int kk = 0; kk < DP / 32; ++kk) {
            const int d = (kk * 4 + g) * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qvalid && d < DH) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * ld + d);
            qf[kk] = v;
        }
        f32x4 st[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DP / 32; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c16) * C::RS + (kk * 4 + gk) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], acc, 0, 0, 0);
            }
            st[t] = acc;
            if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        const int klimit = causal ? (q < N - 1 ? q : N - 1) : N - 1;
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = t * 16 + 4 * g + i;
                const float sv = key <= klimit ? st[t][i] : -3.0e38f;
                st[t][i] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pz = exp2f((st[t][i] - mx) * scale_log2e);
                st[t][i] = pz;
                sum += pz;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        bf16x8 pf[C::KS];
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf[s][i] = (bf16_t)st[2 * s][i];
                if (2 * s + 1 < NT) pf[s][4 + i] = (bf16_t)st[2 * s + 1][i];
                else pf[s][4 + i] = (bf16_t)0.f;
            }
        }
        if (!v_ready) {               // first q-tile of this wave: V must have landed for everybody
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            v_ready = true;
        }
        // ---- O^T = V^T . P^T, V^T fragments by transpose-read: block rows 4g..4g+3 (+16), lane supplies
        // the address of its 4 contiguous values = row (c16>>2), cols 4*(c16&3).. of the [4][16] block
        const char* vlane = Vs + (4 * g + (c16 >> 2)) * C::RS + (c16 & 3) * 8;
#pragma unroll 1
        for (int dt = 0; dt < DP / 16; ++dt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const char* vrow = vlane + dt * 32;
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vrow + s * 32 * C::RS));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vrow + (s * 32 + 16) * C::RS));
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s], acc, 0, 0, 0);
            }
            const int d = dt * 16 + 4 * g;
            if (qvalid && d < DH) {
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(acc[i] * inv);
                *reinterpret_cast<bf16x4*>(out + ((int64_t)b * N + q) * D + h * DH + d) = o;
            }
        }
    }
    if (!v_ready) {   // waves without a q-tile still owe the V barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}


// =================================================================================================
// v3: one PERSISTENT workgroup (9 waves) per frame walking its H heads.  K is double-buffered in LDS and V
// single-buffered (2 x 51 KiB + 54 KiB = 156 KiB of the CU's 160): the LDS-DMA of K(h+1) runs under all of
// head h, the DMA of V(h) under the first S^T of head h, so staging — a third of v2's time, which ran one
// (frame, head) per workgroup with nothing to overlap — disappears behind compute.  9 waves instead of 8:
// the 17 query tiles of 257 tokens become 2 rounds (2,2,...,2,1) instead of 3.
//   per head:  vmcnt(0) | BARRIER A (K(h) landed everywhere, head h-1 finished everywhere)
//              DMA V(h), DMA K(h+1) | S^T + softmax of the first tile | vmcnt(#K pieces) BARRIER B (V(h) landed)
//              P.V ... remaining tiles
// =================================================================================================
// NW waves per workgroup: 9 (17 query tiles = 2,2,...,2,1) or 12 (three waves on every SIMD instead of 3/2/2/2: the busiest SIMD
// still owns 5 tiles, but every SIMD has a third instruction stream to cover softmax VALU and LDS latency with).
__device__ __forceinline__ void sleep_n(int n) {   // ~64 n cycles
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
template <int NW, int CPR> struct ROWS_PER_PIECE_OK { static constexpr bool value = (NW * 64 % CPR == 0) && ((NW * 64 / CPR) % 8 == 0); };
// LEAN (selectable: hirest_attention_select; the default g_attn_variant = 7 is the non-LEAN form + producer wave, v3's bits): the softmax
// arithmetic on fewer VALU instructions, same structure.  rocprofv3 counted 590 VALU
// instructions per 16-query tile against 105 MFMAs (profiles/r03/pmc_summary.md), and the ISA showed where they were: every
// fmaxf operand canonicalised first (v_max_f32 x, x: 68 per tile), 68 scalar v_fma, 68 scalar adds for the row sum.  Now:
//   row max      v_max3_f32 written out (two values per instruction, no canonicalisation: MFMA results are never signalling NaNs)
//   exponents    v_pk_fma_f32, two scores per instruction
//   row sum      no VALU at all when the head dim is padded (88 -> 96): the V image's pad chunk (dims 88..95) is DMA'd from a row
//                of ones, so rows 88..95 of O^T = V^T.P^T ARE sum_k p(k, q) — of the bf16-rounded p the product uses, in the
//                fp32 accumulator — and arrive in the lanes g = 2, 3 of the last d tile; one v_permlane32_swap hands them to
//                g = 0, 1.  (Head dim 64 has no pad: v_pk_add_f32 there.)
// Outputs differ from the former arithmetic in the last bf16 bit of a few elements (another summation order for the row sum).
__device__ __attribute__((aligned(16))) const unsigned short g_attn_ones[8] = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
__device__ __attribute__((aligned(16))) const unsigned short g_attn_zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
// Phase time stamps of workgroup 0 (debug instantiation, hirest_attention_debug_mode bit 8): [step][wave][slot] shader-clock values
constexpr int ATTN_TRACE_STEPS = 64, ATTN_TRACE_WAVES = 12, ATTN_TRACE_SLOTS = 12;
__device__ long long g_attn_trace[ATTN_TRACE_STEPS * ATTN_TRACE_WAVES * ATTN_TRACE_SLOTS];
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// PROD: a tenth wave does nothing but issue the LDS-DMA.  The phase stamps (tools/attn_trace.py, profiles/r04/attention_trace.txt)
// showed every compute wave spending 4 700 - 7 000 of a step's 28 500 cycles INSIDE the 12 global_load_lds instructions it issued
// after barrier A: the CU's DMA path moves ~22 B per clock, the 105 KB burst of V(h) + K(h+1) takes ~4 800 cycles to get
// through it, and a wave whose next instruction is a vector-memory one waits in order for room in that queue.  The producer wave
// takes that wait off the nine compute waves: barrier A | V(h) as one burst, vmcnt(0) | barrier B | K(h+1) paced over the rest
// of the step (one piece per `pace` x 64 cycles, so that the compute waves' own Q loads and output stores find the queue short).
template <int DH, int DP, int NT, bool FAST, bool DBG, int NW = 9, bool LEAN = false, bool PROD = false>   // FAST: not causal and N > 16*(NT-1): only the last key tile holds masked keys
__global__ __launch_bounds__((NW + (PROD ? 1 : 0)) * 64) void attention_kernel_v3(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          int N, int H, float scale_log2e, int causal, int dbg_bits, int nq, int skew,
                                                          int B, int map, int pace, int stagger) {
    const int dbg = DBG ? dbg_bits : 0;   // timing-experiment switches fold away in the production instantiation
    using C = AttnCfg2<DH, DP, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Vs = smem + 2 * C::NPAD * C::RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    // Which (frame, head) pairs this workgroup walks ("steps"; the loop below calls its counter h for the original mapping):
    //   map 0  one frame per workgroup, its H heads in order (grid = B)
    //   map 1  one HEAD per workgroup, frames b0, b0 + bstep, ... (grid = 8 * FL * H persistent workgroups, FL = frames in flight per
    //          XCD): workgroup i sits on XCD i & 7 (round-robin dispatch), and the H * FL workgroups of an XCD work on the H heads
    //          of the same FL frames at the same time.  A head's K / V / Q rows are 176-B slices of 8448-B rows, so every slice
    //          shares its first and last 128-B line with the neighbouring heads': under map 0 the neighbour comes a whole head
    //          step (~13 us, ~6 MB of other traffic through a 4-MB L2) later and the line is fetched again — 3.6 GB fetched per
    //          launch for 2.2 GB of qkv (profiles/r03/pmc_traffic.json); under map 1 the neighbour is another CU of the same XCD
    //          asking within the same few microseconds, an L2 hit.
    int b0, bstep, h0, hstep, nsteps;
    if (map == 0) { b0 = blockIdx.x; bstep = 0; h0 = 0; hstep = 1; nsteps = H; }
    else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, FL = (gridDim.x >> 3) / H;
        h0 = slot % H; hstep = 0; b0 = xcd * FL + slot / H; bstep = 8 * FL;
        nsteps = b0 < B ? (B - b0 + bstep - 1) / bstep : 0;
    }
    if (nsteps == 0) return;
    // The workgroups of a launch start together and do the same work per step: left alone, all 256 CUs ask for their V images in the
    // same microsecond, their K images in the next, and the HBM alternates between saturated and idle.  The first workgroup of every CU
    // waits a different fraction of a step before it starts; those that follow inherit the offset.
    if (stagger > 0 && blockIdx.x < 256) sleep_n(((blockIdx.x >> 3) & 15) * stagger);
    auto sbase = [&](int st) { return qkv + (int64_t)(b0 + st * bstep) * N * ld + (h0 + st * hstep) * DH; };   // q slice of step st
    const int g = lane >> 4, c16 = lane & 15;
    const int gk = g ^ ((-(c16 >> 2)) & 3);
    const int nqt = (nq + 15) >> 4;   // query tiles to compute: all of them, or the leading ones only (hirest_attention_bf16_rows)

    // LDS-DMA piece i of this wave covers 16-B chunks ci = i*64 + lane, i = wave, wave + 9, ...: 576 chunks = 48 rows
    // further each time, same column -> one (row0, column) pair per lane.
    const int ci0 = wave * 64 + lane;
    const int row0 = ci0 / C::CPR, col0 = ci0 - row0 * C::CPR;
    // V image: the 32-B block (16 head dims) at position p of row r holds logical block p ^ ((r >> 2) & 1).  The transpose
    // reads below fetch, per 32-lane half, rows 4g .. 4g + 3 for two values of g at one block position: at the 192-B row
    // stride rows r and r + 4 start 768 B = 3 bank rows apart, i.e. on the SAME 8 banks (2-way conflict on every P.V
    // fragment: 33 % of the kernel's LDS cycles in profiles/r02/pmc_summary.md); the swap puts them 32 B apart.
    constexpr bool VSWAP = C::RS == 192;   // (head dim 64: 128-B rows, a different conflict pattern — image left as it is)
    const int vlog = VSWAP ? ((((col0 >> 1) ^ ((row0 >> 2) & 1)) << 1) | (col0 & 1)) : col0;
    const int vc = vlog * 8 < DH ? vlog : DH / 8 - 1;
    static_assert(ROWS_PER_PIECE_OK<NW, C::CPR>::value, "a lane's rows must keep (row >> 2) & 1 from piece to piece");
    static_assert((NW * 64) % C::CPR == 0, "piece stride must be whole rows");
    constexpr int ROWS_PER_PIECE = NW * 64 / C::CPR;
    auto dma_k = [&](int h, char* dst) -> int {
        int n = 0;
        const bf16_t* src = sbase(h) + D;
        for (int i = wave, row = row0; i < C::NK_INSTR; i += NW, row += ROWS_PER_PIECE) {
            int kc = (col0 & ~3) | ((col0 & 3) ^ ((-(row >> 2)) & 3));      // K image swizzle (see the fragment reads)
            const bool kpad = kc * 8 >= DH;
            kc = kpad ? DH / 8 - 1 : kc;
            const int r = row < N ? row : N - 1;
            const void* from = src + (int64_t)r * ld + kc * 8;
            if (LEAN && DP > DH) from = kpad ? (const void*)g_attn_zeros : from;   // LEAN: the pad chunk is zeros, so Q's pad may hold anything finite
            glds16(from, dst + i * 1024);
            ++n;
        }
        return n;
    };
    constexpr bool SUM_IN_MFMA = LEAN && DP > DH;
    const bool vpad = SUM_IN_MFMA && vlog * 8 >= DH;            // this lane's chunk of every V row is the pad chunk: ones
    auto dma_v = [&](int h) {
        const bf16_t* src = sbase(h) + 2 * D + vc * 8;
        for (int i = wave, row = row0; i < C::NV_INSTR; i += NW, row += ROWS_PER_PIECE) {
            const int r = row < N ? row : N - 1;
            const void* from = src + (int64_t)r * ld;
            if (SUM_IN_MFMA) from = vpad ? (const void*)g_attn_ones : from;
            glds16(from, Vs + i * 1024);
        }
    };
    // Q fragments: lane (c16, g) holds query row qtile*16 + c16, dims 8*(4kk + g)..+7
    auto load_q = [&](int h, int qtile, bf16x8 (&dst)[DP / 32]) {
        const int q = qtile * 16 + c16;
        const bf16_t* base = sbase(h);
        if (LEAN && DP > DH) {
            // unconditional loads from clamped addresses: rows past N repeat row N - 1 (their outputs are never stored), the pad
            // chunk repeats the last real one (it meets the K image's zero chunk)
            const bf16_t* row = base + (int64_t)(q < N ? q : N - 1) * ld;
#pragma unroll
            for (int kk = 0; kk < DP / 32; ++kk) {
                const int d = (kk * 4 + g) * 8;
                dst[kk] = *reinterpret_cast<const bf16x8*>(row + (d < DH ? d : DH - 8));
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < DP / 32; ++kk) {
            const int d = (kk * 4 + g) * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qtile < nqt && q < N && d < DH) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * ld + d);
            dst[kk] = v;
        }
    };

    if (PROD && wave == NW) {
        // piece i = 64 chunks = 64 / CPR rows: a lane's (row, column) pattern repeats every PER = CPR / 4 pieces = 16 rows, which also
        // keeps both image swizzles ((row >> 2) & 3 and (row >> 2) & 1) fixed per lane and pattern slot
        constexpr int PER = C::CPR / 4;
        static_assert(C::NK_INSTR % PER == 0 && C::NV_INSTR % PER == 0, "whole 16-row groups");
        int prow[PER], pkoff[PER], pvoff[PER];
        bool pkpad[PER], pvpad[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int ci = j * 64 + lane, r = ci / C::CPR, c = ci - r * C::CPR;
            prow[j] = r;
            int kc = (c & ~3) | ((c & 3) ^ ((-(r >> 2)) & 3));
            pkpad[j] = kc * 8 >= DH;
            pkoff[j] = (pkpad[j] ? DH / 8 - 1 : kc) * 8;
            const int vl = VSWAP ? ((((c >> 1) ^ ((r >> 2) & 1)) << 1) | (c & 1)) : c;
            pvpad[j] = SUM_IN_MFMA && vl * 8 >= DH;
            pvoff[j] = (vl * 8 < DH ? vl : DH / 8 - 1) * 8;
        }
        auto issue_k = [&](int st, char* dst, int pace_units) {
            const bf16_t* src = sbase(st) + D;
            for (int it0 = 0; it0 < C::NK_INSTR / PER; ++it0) {
                // (head-per-workgroup mapping: the 16 workgroups of a frame would otherwise walk the same rows at the same time)
                const int it = map ? (it0 + h0) % (C::NK_INSTR / PER) : it0;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int row = prow[j] + 16 * it, r = row < N ? row : N - 1;
                    const void* from = src + (int64_t)r * ld + pkoff[j];
                    if (LEAN && DP > DH) from = pkpad[j] ? (const void*)g_attn_zeros : from;
                    glds16(from, dst + (it * PER + j) * 1024);
                    if (pace_units > 0) sleep_n(pace_units);
                }
            }
        };
        auto issue_v = [&](int st) {
            const bf16_t* src = sbase(st) + 2 * D;
            for (int it0 = 0; it0 < C::NV_INSTR / PER; ++it0) {
                const int it = map ? (it0 + h0) % (C::NV_INSTR / PER) : it0;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int row = prow[j] + 16 * it, r = row < N ? row : N - 1;
                    const void* from = src + (int64_t)r * ld + pvoff[j];
                    if (SUM_IN_MFMA) from = pvpad[j] ? (const void*)g_attn_ones : from;
                    glds16(from, Vs + (it * PER + j) * 1024);
                }
            }
        };
        issue_k(0, smem, 0);
        for (int h = 0; h < nsteps; ++h) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(h) landed
            __builtin_amdgcn_s_barrier();                          // A
            issue_v(h);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // V(h) landed
            __builtin_amdgcn_s_barrier();                          // B
            if (h + 1 < nsteps) issue_k(h + 1, smem + ((h + 1) & 1) * (C::NPAD * C::RS), pace);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // A wave owns query tiles `wave` and `wave + 9` (nqt <= 18 on this path).  Their Q fragments for head h+1 are
    // loaded as soon as head h's S^T has consumed the registers, so the global latency never sits in front of a barrier.
    bf16x8 qa[DP / 32], qb[DP / 32];
    load_q(0, wave, qa);
    load_q(0, wave + NW, qb);
    if (!(dbg & 8) && !PROD) dma_k(0, smem);
    for (int h = 0; h < nsteps; ++h) {
        const char* Ks = smem + (h & 1) * (C::NPAD * C::RS);
        auto stamp = [&](int slot) {
            if (DBG && (dbg & 256) && blockIdx.x == 0 && h < ATTN_TRACE_STEPS) {
                const long long t = __builtin_readcyclecounter();
                if (lane == 0) g_attn_trace[(h * ATTN_TRACE_WAVES + wave) * ATTN_TRACE_SLOTS + slot] = t;
            }
        };
        stamp(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(h) pieces and this head's Q fragments (issued a head ago)
#pragma unroll
        for (int kk = 0; kk < DP / 32; ++kk) { asm volatile("" : "+v"(qa[kk])); asm volatile("" : "+v"(qb[kk])); }
        if (!(dbg & 128)) __builtin_amdgcn_s_barrier();        // A: K(h) everywhere, head h-1 done everywhere
        stamp(0);
        if (!(dbg & 8) && !PROD) dma_v(h);
        const int nk_next = (h + 1 < nsteps && !(dbg & 8) && !PROD) ? dma_k(h + 1, smem + ((h + 1) & 1) * (C::NPAD * C::RS)) : 0;
        bool v_ready = false;
        auto tile = [&](int qt, bf16x8 (&qf)[DP / 32], int ord) {
            stamp(1 + 4 * ord);
            const int q = qt * 16 + c16;
            const bool qvalid = q < N;
            f32x4 st[NT];
#pragma unroll
            for (int t = 0; t < NT; t += 2) {   // two key tiles at a time: independent accumulators hide the MFMA latency
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                if (!(dbg & 1))
#pragma unroll
                for (int kk = 0; kk < DP / 32; ++kk) {
                    const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c16) * C::RS + (kk * 4 + gk) * 16);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[kk], a0, 0, 0, 0);
                    if (t + 1 < NT) {
                        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(Ks + ((t + 1) * 16 + c16) * C::RS + (kk * 4 + gk) * 16);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[kk], a1, 0, 0, 0);
                    }
                }
                st[t] = a0;
                if (t + 1 < NT) st[t + 1] = a1;
                if ((t & 2) || !FAST) __builtin_amdgcn_sched_barrier(0);   // FAST: regions of 4 key tiles (12 reads in flight)
            }
            if (h + 1 < nsteps && !(dbg & 32)) load_q(h + 1, qt, qf);            // registers are free: next head's fragments start travelling
            stamp(2 + 4 * ord);   // S^T issued
            // mask (only tiles that can contain masked keys pay for it), row max
            const int klimit = causal ? (q < N - 1 ? q : N - 1) : N - 1;
            float mx = -3.0e38f;
            float sum = 0.f;
            bf16x8 pf[C::KS];
            if (LEAN) {
                float mxb = -3.0e38f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (!FAST || t == NT - 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) st[t][i] = (t * 16 + 4 * g + i) <= klimit ? st[t][i] : -3.0e38f;
                    }
                    if (t & 1) { mxb = max3_raw(mxb, st[t][0], st[t][1]); mxb = max3_raw(mxb, st[t][2], st[t][3]); }
                    else { mx = max3_raw(mx, st[t][0], st[t][1]); mx = max3_raw(mx, st[t][2], st[t][3]); }
                }
                float pa = max3_raw(mx, mxb, mxb), pb = pa;
                lane_swap16(pa, pb); pa = max3_raw(pa, pb, pb); pb = pa; lane_swap32(pa, pb); mx = max3_raw(pa, pb, pb);
                const float nmc = -mx * scale_log2e;
                const f32x2_t sc2 = {scale_log2e, scale_log2e}, nm2 = {nmc, nmc};
                f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < C::KS; ++s2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (2 * s2 + u < NT) {
#pragma unroll
                            for (int ip = 0; ip < 2; ++ip) {
                                const f32x2_t sv = {st[2 * s2 + u][2 * ip], st[2 * s2 + u][2 * ip + 1]};
                                const f32x2_t e = __builtin_elementwise_fma(sv, sc2, nm2);
                                const f32x2_t p2 = (dbg & 2) ? e : f32x2_t{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                                if (!SUM_IN_MFMA) sum2 += p2;
                                pf[s2][4 * u + 2 * ip] = (bf16_t)p2[0];
                                pf[s2][4 * u + 2 * ip + 1] = (bf16_t)p2[1];
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pf[s2][4 + i] = (bf16_t)0.f;
                        }
                    }
                }
                if (!SUM_IN_MFMA) {
                    sum = sum2[0] + sum2[1];
                    float sa = sum, sb = sum; lane_swap16(sa, sb); sum = sa + sb; sa = sum; sb = sum; lane_swap32(sa, sb); sum = sa + sb;
                }
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!FAST || t == NT - 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) st[t][i] = (t * 16 + 4 * g + i) <= klimit ? st[t][i] : -3.0e38f;
                }
                mx = fmaxf(fmaxf(mx, fmaxf(st[t][0], st[t][1])), fmaxf(st[t][2], st[t][3]));
            }
            // (the lane ^ 16 / lane ^ 32 partners through v_permlane16_swap / v_permlane32_swap instead of two ds_bpermute round trips
            //  on the LDS pipe this kernel keeps busy: common.h, wave_max_x)
            { float pa = mx, pb = mx; lane_swap16(pa, pb); mx = fmaxf(pa, pb); pa = mx; pb = mx; lane_swap32(pa, pb); mx = fmaxf(pa, pb); }
            // p = 2^(s*c - mx*c): one fma + one v_exp_f32 per score (arguments are <= 0; underflow flushes to 0)
            const float nmc = -mx * scale_log2e;
#pragma unroll
            for (int s2 = 0; s2 < C::KS; ++s2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e0 = fmaf(st[2 * s2][i], scale_log2e, nmc);
                    const float p0 = (dbg & 2) ? e0 : __builtin_amdgcn_exp2f(e0);
                    sum += p0;
                    pf[s2][i] = (bf16_t)p0;
                    if (2 * s2 + 1 < NT) {
                        const float e1 = fmaf(st[2 * s2 + 1][i], scale_log2e, nmc);
                        const float p1 = (dbg & 2) ? e1 : __builtin_amdgcn_exp2f(e1);
                        sum += p1;
                        pf[s2][4 + i] = (bf16_t)p1;
                    } else {
                        pf[s2][4 + i] = (bf16_t)0.f;
                    }
                }
            }
            { float pa = sum, pb = sum; lane_swap16(pa, pb); sum = pa + pb; pa = sum; pb = sum; lane_swap32(pa, pb); sum = pa + pb; }
            }
            stamp(3 + 4 * ord);   // softmax done
            if (!v_ready) {
                if (!PROD) wait_vm_le(nk_next);      // V(h) pieces are older than the K(h+1) pieces (and the new Q loads)
                if (!(dbg & 64)) __builtin_amdgcn_s_barrier();        // B: V(h) everywhere
                v_ready = true;
            }
            stamp(4 + 4 * ord);   // V ready
            const char* vlane = Vs + (4 * g + (c16 >> 2)) * C::RS + (c16 & 3) * 8;
            // block swap of the V image (see dma_v): even d tiles at +32 (g & 1), odd ones at +32 (1 - (g & 1)) - 32
            const char* vl_e = vlane + ((VSWAP && (g & 1)) ? 32 : 0);
            const char* vl_o = vlane - ((VSWAP && (g & 1)) ? 32 : 0);
            f32x4 oacc[DP / 16];
#pragma unroll
            for (int dt = 0; dt < DP / 16; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(dbg & 4))
#pragma unroll
            for (int s2 = 0; s2 < C::KS; ++s2) {
#pragma unroll
                for (int dt = 0; dt < DP / 16; ++dt) {   // DP/16 independent accumulators per key step
                    const char* vrow = ((dt & 1) ? vl_o : vl_e) + dt * 32 + s2 * 32 * C::RS;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vrow));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vrow + 16 * C::RS));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s2], oacc[dt], 0, 0, 0);
                }
                if ((s2 & 1) || !FAST) __builtin_amdgcn_sched_barrier(0);   // FAST: regions of 2 key steps (24 reads in flight)
            }
            if (SUM_IN_MFMA) {   // rows 88..95 of O^T are the row sums: lanes g = 2, 3 hold them in the last d tile; g = 0, 1 take their partner's
                float sa = oacc[DP / 16 - 1][0], sb = sa;
                lane_swap32(sa, sb);
                sum = sb;
            }
            const float inv = 1.0f / sum;
            // A lane holds 4 consecutive d of every 16-wide d tile (8-B stores, 32 B per query row and instruction).  Two
            // v_permlane16_swap per d-tile pair (semantics: tools/probes/permlane_probe.hip) regroup them so the lane
            // row g owns 8 consecutive d of tile (2*dp + (g&1)): 16-B stores, 64 contiguous bytes per query row.
            static_assert((DP / 16) % 2 == 0 && DH % 8 == 0, "paired d tiles");
#pragma unroll
            for (int dp = 0; dp < DP / 32; ++dp) {
                union { bf16x4 v; unsigned u[2]; } e, o;
#pragma unroll
                for (int i = 0; i < 4; ++i) { e.v[i] = (bf16_t)(oacc[2 * dp][i] * inv); o.v[i] = (bf16_t)(oacc[2 * dp + 1][i] * inv); }
                const auto s0 = __builtin_amdgcn_permlane16_swap(e.u[0], o.u[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(e.u[1], o.u[1], false, false);
                union { bf16x8 v; unsigned u[4]; } w;
                w.u[0] = s0[0]; w.u[1] = s1[0]; w.u[2] = s0[1]; w.u[3] = s1[1];
                const int d = (2 * dp + (g & 1)) * 16 + (g >> 1) * 8;
                if (qvalid && d < DH && !(dbg & 16)) *reinterpret_cast<bf16x8*>(out + ((int64_t)(b0 + h * bstep) * N + q) * D + (h0 + h * hstep) * DH + d) = w.v;
            }
        };
        // De-phasing (hirest_attention_set_skew): all waves leave barrier A together and would run S^T (MFMA), softmax (VALU) and P.V
        // (MFMA) in step, so the second wave of a SIMD competes for the same pipe at every moment and nothing overlaps; parking
        // it for about one S^T phase puts its MFMA phases under the first wave's softmax and vice versa.
        if (skew > 0 && wave >= 4 && wave < 8) sleep_n(skew);
        if (wave < nqt) tile(wave, qa, 0);
        if (wave + NW < nqt) tile(wave + NW, qb, 1);
        stamp(9);
        if (!v_ready) {
            if (!PROD) wait_vm_le(nk_next);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

std::atomic<int> g_attn_dbg{0};   // timing experiments only (hirest_attention_debug_mode)
std::atomic<int> g_attn_skew{12}; // de-phasing of the second wave of each SIMD, in units of ~64 cycles (hirest_attention_set_skew; results unchanged)
std::atomic<int> g_attn_stagger{16};   // start offset between the CUs of an XCD, in units of ~64 cycles x (CU index mod 16) (hirest_attention_set_stagger)
std::atomic<int> g_attn_pace{0};  // producer wave: ~64 x pace cycles between two K pieces (hirest_attention_set_pace)
std::atomic<int> g_attn_map{2};   // hirest_attention_set_mapping: 0 = v3 gives every workgroup one frame (its heads in order), 1 = one head per workgroup over frames when the
                      // shape allows, 2 (default) = automatic: by head below 256 frames — a call of 64 frames is 64 per-frame workgroups on 256 CUs (0.131 ms)
                      // against 256 per-head ones (0.045 ms); from 256 frames on the per-frame walk is as fast or faster (1024: 0.83 vs 1.00 ms).  Same bits.
std::atomic<int> g_attn_variant{7};   // 1 = v1, 2 = v2 (one workgroup per (frame, head)), 3 = v3 (persistent per frame, 9 waves), 4 = v3 with 12 waves,
                          // 5 = v3 with the lean softmax arithmetic, 6 = 5 + producer wave, 7 (default for N > 80) = v3 + producer wave (v3's bits)

template <int DH, int DP, int NT, bool FAST, bool DBG, int NW = 9, bool LEAN = false, bool PROD = false>
int launch3_impl(const bf16_t* qkv, bf16_t* out, int B, int N, int H, float scale, int causal, int nq, hipStream_t s) {
    using C = AttnCfg2<DH, DP, NT>;
    constexpr int LDS = (2 * C::NPAD + C::KP) * C::RS;
    static_assert(LDS <= 163840, "K x2 + V must fit the CU's LDS");
    static HirestDevCfg cfg;
    auto kern = attention_kernel_v3<DH, DP, NT, FAST, DBG, NW, LEAN, PROD>;
    if (int e = hirest_configure(kern, LDS, cfg)) return e;
    // head-per-workgroup mapping when the H heads of FL = 32 / H frames fill the 32 CUs of an XCD exactly (EVA-g/14: H = 16, FL = 2)
    const bool by_head = (g_attn_map == 1 || (g_attn_map == 2 && B < 256)) && H <= 32 && 32 % H == 0 && B >= 8 * (32 / H) * 4;
    const int grid = by_head ? 256 : B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NW + (PROD ? 1 : 0)) * 64), LDS, s, qkv, out, N, H, scale * 1.44269504088896340736f, causal, g_attn_dbg, nq,
                       g_attn_skew, B, by_head ? 1 : 0, g_attn_pace, g_attn_stagger);
    return hirest_launch_status();
}

template <int DH, int DP, int NT, bool FAST>
int launch3(const bf16_t* qkv, bf16_t* out, int B, int N, int H, float scale, int causal, int nq, hipStream_t s) {
    if (g_attn_dbg && FAST && g_attn_variant == 6) return launch3_impl<DH, DP, NT, FAST, true, 9, true, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_variant == 6) return launch3_impl<DH, DP, NT, FAST, false, 9, true, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_dbg && FAST && g_attn_variant == 7) return launch3_impl<DH, DP, NT, FAST, true, 9, false, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_variant == 7) return launch3_impl<DH, DP, NT, FAST, false, 9, false, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_dbg && FAST && g_attn_variant == 5) return launch3_impl<DH, DP, NT, FAST, true, 9, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_dbg && FAST) return launch3_impl<DH, DP, NT, FAST, true>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_variant == 4) return launch3_impl<DH, DP, NT, FAST, false, 12>(qkv, out, B, N, H, scale, causal, nq, s);
    if (g_attn_variant == 5) return launch3_impl<DH, DP, NT, FAST, false, 9, true>(qkv, out, B, N, H, scale, causal, nq, s);
    return launch3_impl<DH, DP, NT, FAST, false>(qkv, out, B, N, H, scale, causal, nq, s);
}


template <int DH, int DP, int NT>
int launch2(const bf16_t* qkv, bf16_t* out, int B, int N, int H, float scale, int causal, hipStream_t s) {
    using C = AttnCfg2<DH, DP, NT>;
    static HirestDevCfg cfg;
    auto kern = attention_kernel_v2<DH, DP, NT>;
    if (int e = hirest_configure(kern, C::LDS_BYTES, cfg)) return e;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(512), C::LDS_BYTES, s, qkv, out, N, H, scale * 1.44269504088896340736f, causal);
    return hirest_launch_status();
}

}  // namespace

extern "C" int hirest_attention_debug_mode(int32_t bits) { g_attn_dbg = bits; return 0; }

extern "C" int hirest_attention_set_skew(int32_t units) {
    if (units < 0 || units > 64) return HIREST_E_BADARG;
    g_attn_skew = units;
    return 0;
}

extern "C" int hirest_attention_debug_trace_read(int64_t* dst, int32_t n) {   // timing experiments: copies the first n stamps
    if (!dst || n <= 0 || n > ATTN_TRACE_STEPS * ATTN_TRACE_WAVES * ATTN_TRACE_SLOTS) return HIREST_E_BADARG;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), sizeof(long long) * n);
}

extern "C" int hirest_attention_set_stagger(int32_t units) {
    if (units < 0 || units > 64) return HIREST_E_BADARG;
    g_attn_stagger = units;
    return 0;
}

extern "C" int hirest_attention_set_pace(int32_t units) {
    if (units < 0 || units > 64) return HIREST_E_BADARG;
    g_attn_pace = units;
    return 0;
}

extern "C" int hirest_attention_set_mapping(int32_t by_head) {
    if (by_head < 0 || by_head > 2) return HIREST_E_BADARG;
    g_attn_map = by_head;
    return 0;
}

extern "C" int hirest_attention_select_kernel(int32_t which) {
    if (which < 1 || which > 7) return HIREST_E_BADARG;
    g_attn_variant = which;
    return 0;
}

extern "C" int hirest_attention_bf16(const hirest_bf16* qkv, hirest_bf16* out, int32_t B, int32_t N, int32_t H,
                                     int32_t dh, float scale, int32_t causal, void* stream) {
    return hirest_attention_bf16_rows(qkv, out, B, N, H, dh, scale, causal, N, stream);
}

extern "C" int hirest_attention_bf16_rows(const hirest_bf16* qkv, hirest_bf16* out, int32_t B, int32_t N, int32_t H,
                                          int32_t dh, float scale, int32_t causal, int32_t q_rows, void* stream) {
    if (!qkv || !out || B <= 0 || N <= 0 || H <= 0 || q_rows <= 0 || q_rows > N) return HIREST_E_BADARG;
    const int nq = q_rows;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bf16_t* q = reinterpret_cast<const bf16_t*>(qkv);
    bf16_t* o = reinterpret_cast<bf16_t*>(out);
    HirestProfScope prof(HIREST_PROF_ATTENTION, causal, (int64_t)B * H, N, dh, s);
    if (g_attn_variant >= 3 && N > 80 && N <= 272 && B >= 64) {
        const bool fast = !causal && N > 256;
        if (dh == 88) return fast ? launch3<88, 96, 17, true>(q, o, B, N, H, scale, causal, nq, s)
                                  : launch3<88, 96, 17, false>(q, o, B, N, H, scale, causal, nq, s);
        if (dh == 64) return fast ? launch3<64, 64, 17, true>(q, o, B, N, H, scale, causal, nq, s)
                                  : launch3<64, 64, 17, false>(q, o, B, N, H, scale, causal, nq, s);
    }
    if (g_attn_variant >= 2) {
        if (dh == 88) {
            if (N <= 80) return launch2<88, 96, 5>(q, o, B, N, H, scale, causal, s);
            if (N <= 272) return launch2<88, 96, 17>(q, o, B, N, H, scale, causal, s);
        } else if (dh == 64) {
            if (N <= 80) return launch2<64, 64, 5>(q, o, B, N, H, scale, causal, s);
            if (N <= 272) return launch2<64, 64, 17>(q, o, B, N, H, scale, causal, s);
        }
        return HIREST_E_SHAPE;
    }
    if (dh == 88) {
        if (N <= 80) return launch<88, 96, 5>(q, o, B, N, H, scale, causal, s);
        if (N <= 272) return launch<88, 96, 17>(q, o, B, N, H, scale, causal, s);
    } else if (dh == 64) {
        if (N <= 80) return launch<64, 64, 5>(q, o, B, N, H, scale, causal, s);
        if (N <= 272) return launch<64, 64, 17>(q, o, B, N, H, scale, causal, s);
    }
    return HIREST_E_SHAPE;
}
