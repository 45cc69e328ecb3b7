// which lane does each cross-lane primitive read?  value = lane id
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)l, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)l, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
    out[256 + l] = dpp<0x128>(l); out[320 + l] = dpp<0x124>(l); out[384 + l] = dpp<0x4E>(l); out[448 + l] = dpp<0xB1>(l);
}
int main() {
    int* d; hipMalloc(&d, 512 * 4); int h[512];
    k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]", "dpp row_ror:8", "dpp row_ror:4", "dpp quad_perm 0x4E", "dpp quad_perm 0xB1"};
    for (int a = 0; a < 8; ++a) { printf("%-20s", names[a]); for (int l = 0; l < 64; ++l) printf(" %d", h[a * 64 + l]); printf("\n"); }
    return 0;
}
