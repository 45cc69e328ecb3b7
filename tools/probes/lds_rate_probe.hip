// What does the LDS deliver to the fragment reads of the bf16 GEMM (gemm.hip pp256_body / pq256), alone and next to what shares the
// CU with them?  (DESIGN 4.1g fitted "7.6 cycles per KB = 135 B/clk, the LDS port" while the counters show the LDS
// array 23 - 29 % active.)  One workgroup per CU, the kernel's own slab image: 2 slots x (256 A rows + 256 W rows) x 128 B, 16-B chunk c of
// row r at position c ^ ((r >> 1) & 7); a wave reads what one 64-deep K step of pq256 reads: 16 A + 8 W fragments of ds_read_b128 (24 KB per
// wave), lgkmcnt(0) once per 12 reads (a phase).  Variants:
//   readers 4 / 8    waves 0-3 only (one reading wave per SIMD) or all eight (two per SIMD)
//   dma              every reading wave also issues the step's LDS-DMA pieces (global_load_lds_dwordx4, 1 KB each; 64 KB per step per CU)
//   mfma sibling     waves 4-7 issue back-to-back v_mfma_f32_16x16x32_bf16 (8 independent accumulators) instead of reading
// Output: LDS bytes returned per core clock per CU (s_memtime over the loop of wave 0) and the MFMA rate of the sibling waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int SLOT = 64 * 1024, WOFF = 32 * 1024;
__device__ __forceinline__ f32x4 lds_read16(const char* p) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p));
    return v;
}
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, float* out, long* t, int iters, int readers, int dma, int mfma_sibling, int one_batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    for (int i = tid; i < 2 * SLOT / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const int frow = lane & 15, fsw = (frow >> 1) & 7, kg = lane >> 4;
    const int a_frag = (grp * 128 + frow) * 128, w_frag = WOFF + (wc * 64 + frow) * 128;
    int koff[2];
    for (int h = 0; h < 2; ++h) koff[h] = ((4 * h + kg) ^ fsw) << 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool reads = wave < readers && !(mfma_sibling && grp == 1);
    const long c0 = clock64(), w0 = wall_clock64();
    if (reads) {
        const char* g = src + (size_t)blockIdx.x * 2 * SLOT + lane * 16;
        for (int it = 0; it < iters; ++it) {
            const char* buf = smem + (it & 1) * SLOT;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {                  // a phase of pq256: one A half (8 reads) + W (4 reads per half, both halves in phase 0)
                f32x4 v[12];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[2 * m + h] = lds_read16(buf + a_frag + (ph * 64 + m * 16) * 128 + koff[h]);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int h = 0; h < 2; ++h) v[8 + 2 * n + h] = lds_read16(buf + w_frag + (ph * 32 + n * 16) * 128 + koff[h]);
                if (dma) {                                    // this wave's share of the step's refill: 64 KB / readers / 2 phases, 1 KB per piece
                    const int pieces = 64 / readers / 2;
                    for (int q = 0; q < pieces; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * 64 + ph * 32 + wave * pieces + q) & 63) * 1024),
                                                         (__attribute__((address_space(3))) void*)(smem + ((it + 1) & 1) * SLOT + ((wave * pieces + q + ph * 32) & 63) * 1024), 16, 0, 0);
                }
                if (!one_batch || ph == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one_batch: all 24 reads of the step behind ONE wait
#pragma unroll
                for (int i = 0; i < 12; ++i) acc += v[i];
            }
            if (dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mfma_sibling && grp == 1) {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)                 // 64 MFMAs per "step", as a wave of pq256 issues
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) acc += c[i];
    }
    const long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) { t[2 * (tid >> 8)] = c1 - c0; t[2 * (tid >> 8) + 1] = w1 - w0; }
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    char* src; float* out; long* t; long h[4];
    hipMalloc(&src, (size_t)cus * 2 * SLOT + 65536); hipMemset(src, 0, (size_t)cus * 2 * SLOT + 65536);
    hipMalloc(&out, (size_t)cus * 512 * 4); hipMalloc(&t, 32);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    const int iters = 4000;
    struct V { const char* name; int readers, dma, sib, one; } vs[] = {
        {"4 reading waves (one per SIMD)", 4, 0, 0}, {"8 reading waves (two per SIMD)", 8, 0, 0},
        {"4 reading waves + their LDS-DMA refill", 4, 1, 0}, {"8 reading waves + their LDS-DMA refill", 8, 1, 0},
        {"4 reading waves, MFMA in the sibling waves", 8, 0, 1}, {"4 reading waves + LDS-DMA, MFMA in the sibling waves", 8, 1, 1},
        {"4 reading waves, 24 reads per wait", 4, 0, 0, 1}, {"4 reading waves, 24 reads per wait, MFMA siblings", 8, 0, 1, 1},
        {"4 reading waves + LDS-DMA, 24 reads per wait, MFMA siblings", 8, 1, 1, 1}};
    printf("# one workgroup per CU on %d CUs, %d steps; per step a reading wave issues 24 ds_read_b128 (24 KB) in two phases\n", cus, iters);
    for (const V& v : vs) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 2 * SLOT, 0, src, out, t, iters, v.readers, v.dma, v.sib, v.one);
        hipDeviceSynchronize(); hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
        const int nread = v.sib ? 4 : v.readers;
        const double bytes = (double)nread * 24 * 1024 * iters, mhz = 100.0 * h[0] / h[1];
        printf("%-56s %6.1f B/clk/CU read (%5.1f cycles per KB)%s  clock %4.0f MHz", v.name, bytes / h[0], h[0] / (bytes / 1024), v.dma ? " + 64 KB/step DMA" : "", mhz);
        if (v.sib) printf("   sibling MFMA duty: %5.1f %% (64 MFMA x 16 cycles per step and wave)", 100.0 * iters * 64 * 16 / h[2]);
        printf("\n");
    }
    return 0;
}
