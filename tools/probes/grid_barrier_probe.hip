// What would "one word of captioning = one launch" cost?  A word step is ~20 dependent phases (caption.hip), each an all-to-all
// hand-off of a [25, 768] fp32 activation between column-parallel GEMMs.  This probe prices the two ways to order such phases on
// this box: (a) one kernel launch per phase on a stream (what ships), (b) a persistent kernel, one 256-thread workgroup per CU,
// with a grid-wide barrier between phases — flat (one device counter + one generation word) and XCD-hierarchical (per-XCD arrival
// counter, the last arriver of an XCD reports to a top counter, the last XCD publishes per-XCD generation words: the form of
// MI355X_MICROARCH.md's "barrier-xcd" row).  The phase body is the hand-off itself: every workgroup reads the 77 KB activation its
// predecessors wrote (L2 / Infinity Cache) and writes its 16-column slice of the next one (write-through stores), i.e. a GEMM phase
// with the weight stream and the MFMAs taken out.  Every spin is bounded: a barrier that does not complete sets a flag and every
// workgroup leaves (no hang).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ROWS = 25, D = 768;
struct Ctl { int xcd_count[8][32]; int top[32]; int gen[8][32]; int flat_count[32]; int flat_gen[32]; int failed[32]; };

__device__ __forceinline__ bool spin_until(const int* word, int target, int* failed) {
    long spins = 0;
    while (__atomic_load_n(word, __ATOMIC_RELAXED) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000 || __atomic_load_n(failed, __ATOMIC_RELAXED)) { __atomic_store_n(failed, 1, __ATOMIC_RELAXED); return false; }
    }
    return true;
}

template <int MODE>   // 0 flat, 1 XCD-hierarchical
__device__ __forceinline__ bool grid_barrier(Ctl* c, int epoch, int nblk, int xcd, int nblk_xcd) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope: this workgroup's stores are visible before it arrives
        if (MODE == 0) {
            const int prev = __hip_atomic_fetch_add(&c->flat_count[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == epoch * nblk - 1) __hip_atomic_store(&c->flat_gen[0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_until(&c->flat_gen[0], epoch, &c->failed[0]);
        } else {
            const int prev = __hip_atomic_fetch_add(&c->xcd_count[xcd][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == epoch * nblk_xcd - 1) {                        // last arriver of this XCD
                const int p2 = __hip_atomic_fetch_add(&c->top[0], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (p2 == epoch * 8 - 1) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) __hip_atomic_store(&c->gen[x][0], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            ok = spin_until(&c->gen[xcd][0], epoch, &c->failed[0]);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok && !__atomic_load_n(&c->failed[0], __ATOMIC_RELAXED);
}

// one hand-off phase: read the whole [ROWS, D] activation `in`, write this workgroup's columns of `out`
__device__ __forceinline__ void phase_body(const float* in, float* out, int blk, int nblk, bool with_body) {
    if (!with_body) return;
    float s = 0.f;
    for (int i = threadIdx.x; i < ROWS * D / 4; i += 256) {
        typedef __attribute__((ext_vector_type(4))) float f4;
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(in) + i);   // (not cached in L1 across phases)
        s += v[0] + v[1] + v[2] + v[3];
    }
    // 3 columns x 25 rows per workgroup (256 workgroups x 3 = 768 columns)
    if (threadIdx.x < ROWS * 3) {
        const int r = threadIdx.x / 3, col = (blk % 256) * 3 + threadIdx.x % 3;
        __hip_atomic_store(out + r * D + col, s * 1e-9f + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(Ctl* c, float* a, float* b, int phases, int epoch0, int with_body) {
    extern __shared__ char pad[];                                       // 100 KB of LDS: one workgroup per CU
    const int blk = blockIdx.x, nblk = gridDim.x, xcd = blk & 7, nblk_xcd = nblk >> 3;
    float* bufs[2] = {a, b};
    for (int p = 0; p < phases; ++p) {
        phase_body(bufs[p & 1], bufs[(p + 1) & 1], blk, nblk, with_body);
        if (!grid_barrier<MODE>(c, epoch0 + p + 1, nblk, xcd, nblk_xcd)) return;
    }
    if (threadIdx.x == 0 && blk == 0) pad[0] = 0;
}

__global__ __launch_bounds__(256) void phase_kernel(const float* in, float* out, int with_body) {
    phase_body(in, out, blockIdx.x, gridDim.x, with_body);
}

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nblk = (cus / 8) * 8;
    Ctl* c; float *a, *b;
    CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&a, ROWS * D * 4)); CK(hipMalloc(&b, ROWS * D * 4));
    CK(hipMemset(a, 0, ROWS * D * 4)); CK(hipMemset(b, 0, ROWS * D * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int PH = 20, WORDS = 48, LDS = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    printf("%d CUs, %d workgroups of 256 threads; a word = %d dependent phases, %d words per measurement\n", cus, nblk, PH, WORDS);
    for (int body = 0; body < 2; ++body) {
        // (a) one launch per phase
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int p = 0; p < PH * WORDS; ++p)
                hipLaunchKernelGGL(phase_kernel, dim3(nblk), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, body);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("body %d  launches:            %7.2f us per phase, %7.1f us per word\n", body, ms * 1e3 / (PH * WORDS), ms * 1e3 / WORDS);
        }
        // (b) persistent kernel, one launch per word (PH barriers) and one launch for all words
        for (int mode = 0; mode < 2; ++mode) {
            for (int whole = 0; whole < 2; ++whole) {
                float best = 1e30f; int failed = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(c, 0, sizeof(Ctl), s));
                    CK(hipEventRecord(e0, s));
                    if (whole) {
                        if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nblk), dim3(256), LDS, s, c, a, b, PH * WORDS, 0, body);
                        else hipLaunchKernelGGL(persistent_kernel<1>, dim3(nblk), dim3(256), LDS, s, c, a, b, PH * WORDS, 0, body);
                    } else {
                        for (int w = 0; w < WORDS; ++w) {
                            if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(nblk), dim3(256), LDS, s, c, a, b, PH, w * PH, body);
                            else hipLaunchKernelGGL(persistent_kernel<1>, dim3(nblk), dim3(256), LDS, s, c, a, b, PH, w * PH, body);
                        }
                    }
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
                    failed |= h.failed[0];
                }
                printf("body %d  %s barrier, %s: %7.2f us per phase, %7.1f us per word%s\n", body, mode ? "xcd " : "flat",
                       whole ? "one launch for 48 words" : "one launch per word    ", best * 1e3 / (PH * WORDS), best * 1e3 / WORDS,
                       failed ? "   [A BARRIER TIMED OUT: numbers invalid]" : "");
            }
        }
    }
    std::vector<float> h(ROWS * D);
    CK(hipMemcpy(h.data(), a, ROWS * D * 4, hipMemcpyDeviceToHost));
    printf("checksum %.3f\n", h[0] + h[ROWS * D - 1]);
    return 0;
}
