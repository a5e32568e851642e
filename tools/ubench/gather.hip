// micro-benchmark: achievable L2-resident row gather rate (rows of LD floats from a small table by random index)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int UNR>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int* __restrict__ idx, long n, int ld, int G, int tile, float* out) {
    const int g = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = 256 / G;
    const long b0 = ((long)blockIdx.x * ngrp + grp) * tile;
    const bool act = g * 4 < ld;
    float4 acc = {0, 0, 0, 0};
    for (long q0 = b0; q0 < b0 + tile && q0 < n; q0 += UNR) {
        float4 rv[UNR];
#pragma unroll
        for (int e = 0; e < UNR; ++e) if (q0 + e < n && act) rv[e] = *reinterpret_cast<const float4*>(tab + (size_t)idx[q0 + e] * ld + g * 4);
#pragma unroll
        for (int e = 0; e < UNR; ++e) if (q0 + e < n && act) { acc.x += rv[e].x; acc.y += rv[e].y; acc.z += rv[e].z; acc.w += rv[e].w; }
    }
    if (act && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 100, nrows = argc > 2 ? atoi(argv[2]) : 3952;
    const long n = 939809; const int G = 32, tile = 64;
    std::vector<float> t((size_t)nrows * ld, 1.f); std::vector<int> ix(n);
    srand(1); for (long i = 0; i < n; ++i) ix[i] = rand() % nrows;
    float *dt, *dout; int* di;
    hipMalloc(&dt, t.size() * 4); hipMalloc(&di, n * 4); hipMalloc(&dout, 4);
    hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice); hipMemcpy(di, ix.data(), n * 4, hipMemcpyHostToDevice);
    const int ngrp = 256 / G; const int grid = (int)((n + (long)ngrp * tile - 1) / ((long)ngrp * tile));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int unr : {4, 8, 16}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            for (int it = 0; it < 20; ++it) {
                if (unr == 4) hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, dt, di, n, ld, G, tile, dout);
                else if (unr == 8) hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, dt, di, n, ld, G, tile, dout);
                else hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(256), 0, 0, dt, di, n, ld, G, tile, dout);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("ld=%d rows=%d UNR=%2d: %.1f us per pass, %.2f TB/s (useful %d B/row)\n", ld, nrows, unr, 1e3 * ms / 20, n * (double)(ld * 4) / (ms / 20 * 1e-3) / 1e12, ld * 4);
        }
    }
    return 0;
}
