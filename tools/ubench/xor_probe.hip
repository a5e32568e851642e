// probe: lane^4 / lane^8 via ds_swizzle (bit mode), lane^16 via v_permlane16_swap, lane^32 via v_permlane32_swap -- against __shfl_xor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int lane = threadIdx.x, v = 1000 + lane * 7;
    const int x4 = __builtin_amdgcn_ds_swizzle(v, 0x101F), x8 = __builtin_amdgcn_ds_swizzle(v, 0x201F);
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const int x16 = (lane & 16) ? r[0] : r[1];
    auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    const int x32 = (lane & 32) ? q[0] : q[1];
    o[lane] = (x4 == __shfl_xor(v, 4)) | ((x8 == __shfl_xor(v, 8)) << 1) | ((x16 == __shfl_xor(v, 16)) << 2) | ((x32 == __shfl_xor(v, 32)) << 3);
}
int main() {
    int* d; int h[64];
    hipMalloc(&d, 256); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int all = 15; for (int i = 0; i < 64; ++i) all &= h[i];
    printf("xor4 %d xor8 %d xor16 %d xor32 %d\n", all & 1, (all >> 1) & 1, (all >> 2) & 1, (all >> 3) & 1);
    return 0;
}
