// micro-benchmark (VERDICT r02 item 7 / SURVEY 8 f3, "blocked-user MFMA V-step"): the rating block of the NB densest users of a
// shape as DENSE fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), at the block's real density:
//   dense SDDMM   B = U_blk P^T (NB x d2, K = r), then b[rating] picked out of the tiles through a static (tile, cell) -> rating map
//                 (replaces k_sddmm's row gathers for those users: pcrpp.cpp:266-271)
//   dense SpMM    O = C^T U_blk (d2 x r, K = NB), C scattered into tiles from c[rating] through the same kind of map
//                 (replaces k_spmm's row gathers: pcrpp.cpp:323-327)
// Both are checked against a host computation over the block's ratings.  The per-user rating counts come from a file (one count
// per line: the real counts of the shape's longest users, written by tools/exp_vblock.py); item sets are uniform without
// replacement, as the shapes' generator draws them.
// usage: vblock_probe <counts file> <d2> <r> [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));

// cell of accumulator element v of lane l in a 32 x 32 tile: row (A side) and column (B side)
__host__ __device__ inline int cell_row(int v, int l) { return 8 * (v / 4) + 4 * (l / 32) + (v % 4); }
__host__ __device__ inline int cell_col(int l) { return l % 32; }

// ---- dense SDDMM: one wave per (user group, item group) tile; the workgroup's 4 waves share the user group's rows in LDS
constexpr int LSTR = 101;      // LDS row stride in floats (odd: the 32 lanes of a half-wave read 32 rows at one k conflict-free)
__global__ __launch_bounds__(256) void k_dense_sddmm(const float* __restrict__ U, const float* __restrict__ P, const int* __restrict__ map,
                                                     float* __restrict__ out, int nig, int ld, int r) {
    __shared__ float A[32 * LSTR];
    __shared__ float B[4][32 * LSTR];
    const int ug = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int ig = blockIdx.x * 4 + w;
    for (int t = threadIdx.x; t < 32 * ld; t += 256) A[(t / ld) * LSTR + t % ld] = U[(size_t)ug * 32 * ld + t];
    if (ig < nig)
        for (int t = l; t < 32 * ld; t += 64) B[w][(t / ld) * LSTR + t % ld] = P[(size_t)ig * 32 * ld + t];
    __syncthreads();
    if (ig >= nig) return;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* a = A + (l % 32) * LSTR + l / 32;
    const float* b = B[w] + (l % 32) * LSTR + l / 32;
    for (int k = 0; k < r; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    const int* m = map + ((size_t)ug * nig + ig) * 1024;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int d = m[v * 64 + l];
        if (d >= 0) out[d] = acc[v];
    }
}

// ---- dense SpMM: one wave per (item group, 32 rank columns); K runs over all users of the block
__global__ __launch_bounds__(256) void k_dense_spmm(const float* __restrict__ c, const int* __restrict__ map2, const float* __restrict__ U,
                                                    float* __restrict__ out, int nug, int nig, int ld) {
    const int ig = blockIdx.x, t = threadIdx.x >> 6, l = threadIdx.x & 63;       // t: column tile (4 x 32 >= ld)
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int col = 32 * t + l % 32;
    for (int ug = 0; ug < nug; ++ug) {
        const int* m = map2 + ((size_t)ug * nig + ig) * 1024;
        const float* Ub = U + (size_t)ug * 32 * ld;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int d = m[s * 64 + l];                                         // cell (user 2 s + l / 32, item l % 32)
            const float a = d >= 0 ? c[d] : 0.f;
            const float b = col < ld ? Ub[(size_t)(2 * s + l / 32) * ld + col] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    if (col < ld) {
#pragma unroll
        for (int v = 0; v < 16; ++v) out[((size_t)ig * 32 + cell_row(v, l)) * ld + col] = acc[v];
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: vblock_probe <counts file> <d2> <r> [reps]\n"); return 1; }
    const int d2 = atoi(argv[2]), r = atoi(argv[3]), reps = argc > 4 ? atoi(argv[4]) : 50;
    std::vector<int> cnt;
    { FILE* f = fopen(argv[1], "r"); int x; while (f && fscanf(f, "%d", &x) == 1) cnt.push_back(std::min(x, d2)); if (f) fclose(f); }
    const int nb = (int)cnt.size() / 32 * 32;
    if (nb == 0 || r % 4 || r > 100) { printf("need a multiple of 32 users and r %% 4 == 0, r <= 100\n"); return 1; }
    cnt.resize(nb);
    const int ld = r, nug = nb / 32, nig = (d2 + 31) / 32, d2p = nig * 32;
    std::mt19937_64 rng(7);
    // ratings of the block: per user a uniform item set without replacement, ascending (CSR order of the block)
    std::vector<int64_t> uptr(nb + 1, 0);
    std::vector<int> item;
    std::vector<char> mark(d2);
    for (int u = 0; u < nb; ++u) {
        std::fill(mark.begin(), mark.end(), 0);
        for (int got = 0; got < cnt[u];) { const int j = (int)(rng() % d2); if (!mark[j]) { mark[j] = 1; ++got; } }
        for (int j = 0; j < d2; ++j) if (mark[j]) item.push_back(j);
        uptr[u + 1] = (int64_t)item.size();
    }
    const int64_t nnz = (int64_t)item.size();
    std::vector<int> map((size_t)nug * nig * 1024, -1), map2((size_t)nug * nig * 1024, -1);
    for (int u = 0; u < nb; ++u)
        for (int64_t z = uptr[u]; z < uptr[u + 1]; ++z) {
            const int j = item[z], ug = u / 32, ig = j / 32, ru = u % 32, cj = j % 32;
            // map: [v][l] with cell_row(v, l) == ru, cell_col(l) == cj  ->  l = cj + 32 * ((ru % 8) / 4), v = 4 * (ru / 8) + ru % 4
            const int l = cj + 32 * ((ru % 8) / 4), v = 4 * (ru / 8) + ru % 4;
            map[((size_t)ug * nig + ig) * 1024 + v * 64 + l] = (int)z;
            // map2: [s][l] with user 2 s + l / 32 == ru, item l % 32 == cj
            map2[((size_t)ug * nig + ig) * 1024 + (ru / 2) * 64 + (ru % 2) * 32 + cj] = (int)z;
        }
    std::vector<float> U((size_t)nb * ld), P((size_t)d2p * ld, 0.f), cv(nnz);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& x : U) x = nd(rng);
    for (size_t i = 0; i < (size_t)d2 * ld; ++i) P[i] = nd(rng);
    for (auto& x : cv) x = nd(rng);
    float *dU, *dP, *dc, *dout, *dO; int *dmap, *dmap2;
    CK(hipMalloc(&dU, U.size() * 4)); CK(hipMalloc(&dP, P.size() * 4)); CK(hipMalloc(&dc, cv.size() * 4)); CK(hipMalloc(&dout, nnz * 4));
    CK(hipMalloc(&dO, (size_t)d2p * ld * 4)); CK(hipMalloc(&dmap, map.size() * 4)); CK(hipMalloc(&dmap2, map2.size() * 4));
    CK(hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, cv.data(), cv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmap, map.data(), map.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dmap2, map2.data(), map2.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3 * ms / reps;
    };
    const double us_sd = time_it([&]() { hipLaunchKernelGGL(k_dense_sddmm, dim3((nig + 3) / 4, nug), dim3(256), 0, 0, dU, dP, dmap, dout, nig, ld, r); });
    const double us_sp = time_it([&]() { hipLaunchKernelGGL(k_dense_spmm, dim3(nig), dim3(256), 0, 0, dc, dmap2, dU, dO, nug, nig, ld); });
    CK(hipGetLastError());
    // ---- check against the host
    std::vector<float> out(nnz), O((size_t)d2p * ld);
    CK(hipMemcpy(out.data(), dout, nnz * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double e_sd = 0, e_sp = 0, n_sd = 0, n_sp = 0;
    std::vector<double> Oh((size_t)d2 * ld, 0.0);
    for (int u = 0; u < nb; ++u)
        for (int64_t z = uptr[u]; z < uptr[u + 1]; ++z) {
            double s = 0;
            for (int k = 0; k < r; ++k) s += (double)U[(size_t)u * ld + k] * P[(size_t)item[z] * ld + k];
            e_sd = std::max(e_sd, std::fabs(s - out[z])); n_sd = std::max(n_sd, std::fabs(s));
            for (int k = 0; k < r; ++k) Oh[(size_t)item[z] * ld + k] += (double)cv[z] * U[(size_t)u * ld + k];
        }
    for (size_t i = 0; i < Oh.size(); ++i) { e_sp = std::max(e_sp, std::fabs(Oh[i] - O[i])); n_sp = std::max(n_sp, std::fabs(Oh[i])); }
    const double dens = (double)nnz / ((double)nb * d2);
    printf("block: %d users x %d items, %lld ratings, density %.1f %%, r = %d\n", nb, d2, (long long)nnz, 100 * dens, r);
    printf("dense SDDMM (MFMA 32x32x2 f32): %8.2f us = %.4f ns per rating, %.1f TFLOP/s   max rel err %.2e\n", us_sd, 1e3 * us_sd / nnz,
           2.0 * nb * d2p * r / us_sd * 1e-6, e_sd / n_sd);
    printf("dense SpMM  (MFMA 32x32x2 f32): %8.2f us = %.4f ns per rating, %.1f TFLOP/s   max rel err %.2e\n", us_sp, 1e3 * us_sp / nnz,
           2.0 * nb * d2p * 128 / us_sp * 1e-6, e_sp / n_sp);
    return (e_sd / n_sd < 1e-4 && e_sp / n_sp < 1e-4) ? 0 : 2;
}
