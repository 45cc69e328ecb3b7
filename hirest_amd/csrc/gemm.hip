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
    int nbm, nbn, ppx;   // tile counts, M-panels per XCD
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int GROUP_M = 8;                        // M-panels walked together inside one XCD

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
    } else {  // HIREST_EPI_PATCH_POS_F32
        const int b = m / p.P, pp = m - b * p.P;
        const int64_t orow = (int64_t)b * (p.P + 1) + 1 + pp;
        f32x4 ps = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(1 + pp) * p.N + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = v + ps;
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_t128(GemmP p) {
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
int launch(const GemmP& p, hipStream_t s) {
    const int grid = 8 * p.ppx * p.nbn;
    hipLaunchKernelGGL(gemm_t128<EPI>, dim3(grid), dim3(256), 2 * STAGE_BYTES, s, p);
    return hirest_launch_status();
}

}  // namespace

extern "C" int hirest_gemm_bf16(const hirest_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->W || !a->out) return HIREST_E_BADARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return HIREST_E_BADARG;
    if (a->K % BK != 0 || a->N % 4 != 0 || a->lda % 8 != 0 || a->ldw % 8 != 0) return HIREST_E_SHAPE;
    GemmP p;
    p.A = reinterpret_cast<const bf16_t*>(a->A); p.lda = a->lda;
    p.W = reinterpret_cast<const bf16_t*>(a->W); p.ldw = a->ldw;
    p.bias = a->bias; p.out = a->out; p.ldo = a->ldo;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.pos = a->pos; p.P = a->patches_per_frame;
    p.nbm = (a->M + BM - 1) / BM; p.nbn = (a->N + BN - 1) / BN;
    p.ppx = (p.nbm + 7) / 8;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HirestProfScope prof(HIREST_PROF_GEMM, a->epilogue, a->M, a->N, a->K, s);
    switch (a->epilogue) {
        case HIREST_EPI_BIAS_BF16: return launch<HIREST_EPI_BIAS_BF16>(p, s);
        case HIREST_EPI_BIAS_GELU_BF16: return launch<HIREST_EPI_BIAS_GELU_BF16>(p, s);
        case HIREST_EPI_BIAS_QGELU_BF16: return launch<HIREST_EPI_BIAS_QGELU_BF16>(p, s);
        case HIREST_EPI_BIAS_RESID_F32: return launch<HIREST_EPI_BIAS_RESID_F32>(p, s);
        case HIREST_EPI_BIAS_F32: return launch<HIREST_EPI_BIAS_F32>(p, s);
        case HIREST_EPI_PATCH_POS_F32:
            if (!a->pos || a->patches_per_frame <= 0) return HIREST_E_BADARG;
            return launch<HIREST_EPI_PATCH_POS_F32>(p, s);
        default: return HIREST_E_BADARG;
    }
}
