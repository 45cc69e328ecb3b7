// Probe: per-CU LDS-DMA (global_load_lds_dwordx4) streaming rate from an L2/MALL-resident matrix,
// comparing 64-B row segments (16 rows per wave-instruction) with 128-B segments (8 rows), at the GEMM's
// access pattern: every block walks K over its own 256-row panel; consecutive blocks share panels like GEMM tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %d\n", (int)e, __LINE__); exit(1);} } while (0)

template <int SEG>   // bytes per row segment: 64 or 128
__global__ __launch_bounds__(512) void dma_stream(const char* __restrict__ A, int64_t row_bytes, int rows_total, int ksteps, int depth, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> panel of 512 rows (256 "A" + 256 "W"), neighbours share half like GEMM tiles on one XCD
    const int panel = (blockIdx.x >> 3) % (rows_total / 512);
    const char* base = A + (int64_t)panel * 512 * row_bytes;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per instruction
    constexpr int PIECES = 512 / RPI;        // instructions per k-step for 512 rows
    constexpr int PPW = PIECES / 8;          // per wave
    int issued = 0;
    for (int k = 0; k < ksteps; ++k) {
        char* slot = smem + (k & 3) * (512 * SEG);
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = wave * PPW + q;
            const int row = piece * RPI + lane / LPR;
            const char* src = base + (int64_t)row * row_bytes + (int64_t)(k % (int)(row_bytes / SEG)) * SEG + (lane % LPR) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
        }
        issued += PPW;
        // keep `depth` k-steps in flight
        if (depth == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (SEG == 64) { if (depth == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else { if (depth == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[issued & 1023];
}

// Same access pattern through the register path: global_load_dwordx4 -> VGPR -> ds_write_b128, one k-step of loads in
// flight while the previous one is written to LDS.  NWAVES = 8 (8 loads per wave per step) or 4 (16 loads).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void vgpr_stream(const char* __restrict__ A, int64_t row_bytes, int rows_total, int ksteps, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int panel = (blockIdx.x >> 3) % (rows_total / 512);
    const char* base = A + (int64_t)panel * 512 * row_bytes;
    constexpr int PPW = 64 / NWAVES;
    u32x4 cur[PPW], nxt[PPW];
    auto load = [&](int k, u32x4 (&dst)[PPW]) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = wave * PPW + q;
            const int row = piece * 8 + lane / 8;
            const char* src = base + (int64_t)row * row_bytes + (int64_t)(k % (int)(row_bytes / 128)) * 128 + (lane % 8) * 16;
            dst[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
        }
    };
    load(0, cur);
    for (int k = 0; k < ksteps; ++k) {
        load(k + 1, nxt);
        char* slot = smem + (k & 1) * 65536;
#pragma unroll
        for (int q = 0; q < PPW; ++q) *reinterpret_cast<u32x4*>(slot + (wave * PPW + q) * 1024 + lane * 16) = cur[q];
#pragma unroll
        for (int q = 0; q < PPW; ++q) cur[q] = nxt[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[ksteps & 1023];
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 65536;        // 65536 rows = 184 MB (Infinity-Cache resident); 4096 = 11.5 MB (L2)
    const int64_t row_bytes = 2816;   // K=1408 bf16
    char* A; CK(hipMalloc(&A, rows * row_bytes)); CK(hipMemset(A, 1, rows * row_bytes));
    int* sink; CK(hipMalloc(&sink, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)dma_stream<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)dma_stream<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 * 2 > 163840 ? 163840 : 262144));
    for (int depth = 1; depth <= 3; ++depth)
    for (int seg = 64; seg <= 128; seg += 64) {
        if (seg == 128 && depth > 2) continue;   // 4 x 64 KiB does not fit; slots alias (timing only)
        const int ksteps = seg == 64 ? 44 * 8 : 22 * 8;
        const int blocks = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (seg == 64) hipLaunchKernelGGL(dma_stream<64>, dim3(blocks), dim3(512), 131072, 0, A, row_bytes, rows, ksteps, depth, sink);
            else hipLaunchKernelGGL(dma_stream<128>, dim3(blocks), dim3(512), 163840, 0, A, row_bytes, rows, ksteps, depth, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = (double)blocks * ksteps * 512 * seg;
            if (rep) printf("seg %3d B, %d k-steps in flight: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per CU\n", seg, depth, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
        }
    }
    CK(hipFuncSetAttribute((const void*)vgpr_stream<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)vgpr_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int nw = 4; nw >= 4; nw -= 4) {   // (the 8-wave instantiation is kept for experiments; its loop gets optimised away)
        const int ksteps = 22 * 8, blocks = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (nw == 8) hipLaunchKernelGGL(vgpr_stream<8>, dim3(blocks), dim3(512), 131072, 0, A, row_bytes, rows, ksteps, sink);
            else hipLaunchKernelGGL(vgpr_stream<4>, dim3(blocks), dim3(256), 131072, 0, A, row_bytes, rows, ksteps, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = (double)blocks * ksteps * 512 * 128;
            if (rep) printf("register path, %d waves, 128-B segments, 1 k-step in flight: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per CU\n", nw, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
        }
    }
    return 0;
}
