// micro-benchmark: achievable L2-resident row gather rate and the cost of each SDDMM ingredient
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ float reduce8(const float (&a)[8], int g, int G) {
    float b[4], c[2], d; const bool b0 = g & 1, b1 = g & 2, b2 = g & 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { float s = b0 ? a[i] : a[i + 4], k = b0 ? a[i + 4] : a[i]; b[i] = k + __shfl_xor(s, 1); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { float s = b1 ? b[i] : b[i + 2], k = b1 ? b[i + 2] : b[i]; c[i] = k + __shfl_xor(s, 2); }
    { float s = b2 ? c[0] : c[1], k = b2 ? c[1] : c[0]; d = k + __shfl_xor(s, 4); }
    if (G > 8) d += __shfl_xor(d, 8);
    if (G > 16) d += __shfl_xor(d, 16);
    return d;
}
// MODE 0: raw gather, ids from global; 1: ids staged in LDS; 2: + dot with a register vector + reduce8 + one output per row
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int* __restrict__ idx, long n, int ld, int G, int tile, float* out) {
    extern __shared__ int s_row[];
    const int g = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = 256 / G;
    const int span = ngrp * tile;
    const long b0 = (long)blockIdx.x * span;
    const int nb = (int)((n - b0 < span) ? (n - b0) : span);
    if (MODE >= 1) { for (int t = threadIdx.x; t < nb; t += 256) s_row[t] = idx[b0 + t]; __syncthreads(); }
    const bool act = g * 4 < ld; const int chv = act ? g : 0;
    float4 acc = {0, 0, 0, 0};
    float4 uv = act ? *reinterpret_cast<const float4*>(tab + chv * 4) : float4{0, 0, 0, 0};
    const int l0 = grp * tile, l1 = (l0 + tile < nb) ? l0 + tile : nb;
    const int rho = 4 * (g & 1) + (g & 2) + ((g >> 2) & 1);
    for (int q0 = l0; q0 + 8 <= l1; q0 += 8) {
        float4 rv[8]; int ri[8];
        if (MODE >= 1) { int4 i0 = *reinterpret_cast<const int4*>(s_row + q0), i1 = *reinterpret_cast<const int4*>(s_row + q0 + 4);
                         ri[0]=i0.x; ri[1]=i0.y; ri[2]=i0.z; ri[3]=i0.w; ri[4]=i1.x; ri[5]=i1.y; ri[6]=i1.z; ri[7]=i1.w; }
        else { for (int e = 0; e < 8; ++e) ri[e] = idx[b0 + q0 + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = *reinterpret_cast<const float4*>(tab + (size_t)ri[e] * ld + chv * 4);
        if (MODE == 2) {
            float part[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) part[e] = rv[e].x * uv.x + rv[e].y * uv.y + rv[e].z * uv.z + rv[e].w * uv.w;
            const float tot = reduce8(part, g, G);
            if (g < 8) out[b0 + q0 + rho] = tot;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc.x += rv[e].x; acc.y += rv[e].y; acc.z += rv[e].z; acc.w += rv[e].w; }
        }
    }
    if (MODE < 2 && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_touch(float* t, long n) { long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) t[i] = t[i] * 1.0f; }
int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 100, nrows = argc > 2 ? atoi(argv[2]) : 3952, tile = argc > 3 ? atoi(argv[3]) : 64;
    const long n = 939809 / 512 * 512; const int G = 32;
    std::vector<float> t((size_t)nrows * ld, 1.f); std::vector<int> ix(n);
    srand(1); for (long i = 0; i < n; ++i) ix[i] = rand() % nrows;
    float *dt, *dout; int* di;
    CK(hipMalloc(&dt, t.size() * 4 + 4096)); CK(hipMalloc(&di, n * 4)); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(di, ix.data(), n * 4, hipMemcpyHostToDevice));
    const int ngrp = 256 / G; const int span = ngrp * tile; const int grid = (int)((n + span - 1) / span);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            for (int it = 0; it < 20; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), span * 4, 0, dt, di, n, ld, G, tile, dout);
                else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), span * 4, 0, dt, di, n, ld, G, tile, dout);
                else hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), span * 4, 0, dt, di, n, ld, G, tile, dout);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("ld=%d rows=%d tile=%d mode=%d: %.1f us per pass, %.2f TB/s\n", ld, nrows, tile, mode, 1e3 * ms / 20, n * (double)(ld * 4) / (ms / 20 * 1e-3) / 1e12);
    }
    {   // table rewritten by another kernel before every pass (as the CG direction is): first touches miss the per-XCD L2
        double tot = 0; long tn = (long)nrows * ld;
        for (int it = 0; it < 20; ++it) {
            hipLaunchKernelGGL(k_touch, dim3((tn + 255) / 256), dim3(256), 0, 0, dt, tn);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), span * 4, 0, dt, di, n, ld, G, tile, dout);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms;
        }
        printf("mode=2 with the table rewritten before each pass: %.1f us per pass\n", 1e3 * tot / 20);
    }
    return 0;
}
