// probe: wave-wide inclusive scan of doubles with DPP row_shr 1/2/4/8 + row_bcast15 + row_bcast31 against a serial scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_d(double v) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_d<0x111, 0xF>(v);      // row_shr:1
    v += dpp_d<0x112, 0xF>(v);      // row_shr:2
    v += dpp_d<0x114, 0xF>(v);      // row_shr:4
    v += dpp_d<0x118, 0xF>(v);      // row_shr:8
    v += dpp_d<0x142, 0xA>(v);      // row_bcast:15 -> rows 1, 3
    v += dpp_d<0x143, 0xC>(v);      // row_bcast:31 -> rows 2, 3
    return v;
}
__global__ void k(const double* in, double* out) { out[threadIdx.x] = wave_incl_scan(in[threadIdx.x]); }
int main() {
    double h[64], r[64], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = sin(i * 1.7) * 3 + i * 0.01;
    hipMalloc(&di, 512); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(r, dout, 512, hipMemcpyDeviceToHost);
    double s = 0, worst = 0;
    for (int i = 0; i < 64; ++i) { s += h[i]; worst = fmax(worst, fabs(r[i] - s)); }
    printf("max abs diff %.3e (last %.6f vs %.6f)\n", worst, r[63], s);
    return 0;
}
