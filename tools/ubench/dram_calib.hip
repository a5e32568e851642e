// Dev tool (GPU box): what do the memory-side TCC counters of rocprofv3 count on gfx950?
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_build/dram_calib tools/ubench/dram_calib.hip
//     rocprofv3 --pmc <counter> --output-format csv -d <dir> -- tools/ubench/_build/dram_calib      (one counter group per pass)
// Known byte counts in the access patterns of the product's kernels (16-byte lanes, whole-row gathers), one kernel symbol per
// pattern so that a per-dispatch PMC table separates them (tools/pmc_dram_calib.py):
//   k_calib_read      1 GiB streamed once (4 x the 256 MiB Infinity Cache): every byte must come from HBM
//   k_calib_write     1 GiB streamed out
//   k_calib_copy      1 GiB -> 1 GiB
//   k_calib_reread    a 64 MiB buffer read once per launch, 8 launches back to back: launches 2..8 can be served by the
//                     Infinity Cache -- a counter that excludes Infinity-Cache hits drops to ~0 there, one that counts
//                     L2 <-> fabric requests does not
//   k_calib_l2read    a 2 MiB buffer read 32 times inside one launch by every XCD: the L2 serves it (8 x 2 MiB compulsory)
//   k_calib_gather    whole 400-byte rows (the ml1m fp32 factor row) at uniformly random positions: tables of 1.5 MB (L2),
//                     7 MB, 109 MB (Infinity Cache) and 1 GiB (HBM), 4 M rows per launch, 3 launches each
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_calib_read(const float4 *__restrict__ a, size_t n, float *out)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(256) k_calib_write(float4 *__restrict__ a, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_float4(v, v, v, v);
}

__global__ void __launch_bounds__(256) k_calib_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

__global__ void __launch_bounds__(256) k_calib_reread(const float4 *__restrict__ a, size_t n, float *out)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(256) k_calib_l2read(const float4 *__restrict__ a, size_t n, int reps, float *out)
{
    float s = 0.f;
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            float4 v = a[(i + (size_t)r * 4099) % n];
            s += v.x + v.y + v.z + v.w;
        }
    if (s == 123.456f) out[0] = s;
}

// one 32-lane group per row, 25 lanes x 16 B = 400 B; rows at hashed positions of a table of `rows` rows
template <int TAG>
__global__ void __launch_bounds__(256) k_calib_gather(const float *__restrict__ table, uint32_t rows, uint32_t n_gather, uint32_t seed, float *out)
{
    const int lane = threadIdx.x & 31;
    const size_t grp = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 5, ngrp = ((size_t)gridDim.x * blockDim.x) >> 5;
    float s = 0.f;
    for (size_t g = grp; g < n_gather; g += ngrp) {
        uint32_t h = (uint32_t)g * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float4 *row = reinterpret_cast<const float4 *>(table + (size_t)(h % rows) * 100);
        if (lane < 25) {
            float4 v = row[lane];
            s += v.x + v.y + v.z + v.w;
        }
    }
    if (s == 123.456f) out[0] = s;
}

template <int TAG>
static void gather(const float *table, size_t table_bytes, float *out)
{
    uint32_t rows = (uint32_t)(table_bytes / 400);
    for (int rep = 0; rep < 3; ++rep)
        k_calib_gather<TAG><<<4096, 256>>>(table, rows, 4u << 20, 77u + rep, out);
    CK(hipDeviceSynchronize());
    printf("gather<%d>: table %.1f MB (%u rows of 400 B), 4194304 rows = %.1f MB per launch, 3 launches\n", TAG, table_bytes / 1e6, rows,
           (4u << 20) * 400.0 / 1e6);
}

// dram_calib loop <pattern> <seconds>: the pattern launched back to back for that long (for a sampler of the memory controllers'
// activity beside it: tools/umc_sample.py) -- prints the byte rate the pattern's known bytes give
static int loop_mode(const char* pat, double seconds)
{
    const size_t GiB = 1ull << 30;
    float4 *a, *b;
    float *out;
    CK(hipMalloc(&a, GiB)); CK(hipMalloc(&b, GiB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, GiB)); CK(hipMemset(b, 0, GiB)); CK(hipDeviceSynchronize());
    const size_t n = GiB / 16;
    const float* t = reinterpret_cast<const float*>(b);
    double bytes = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    long launches = 0;
    const double t0 = (double)clock() / CLOCKS_PER_SEC;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto elapsed = [&]() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (ts.tv_sec - ts0.tv_sec) + 1e-9 * (ts.tv_nsec - ts0.tv_nsec); };
    (void)t0;
    while (elapsed() < seconds) {
        for (int rep = 0; rep < 16; ++rep, ++launches) {
            if (!strcmp(pat, "read")) { k_calib_read<<<8192, 256>>>(a, n, out); bytes += GiB; }
            else if (!strcmp(pat, "write")) { k_calib_write<<<8192, 256>>>(b, n, 1.f); bytes += GiB; }
            else if (!strcmp(pat, "copy")) { k_calib_copy<<<8192, 256>>>(a, b, n); bytes += 2.0 * GiB; }
            else if (!strcmp(pat, "reread")) { k_calib_reread<<<8192, 256>>>(a, (64ull << 20) / 16, out); bytes += 64ull << 20; }
            else if (!strcmp(pat, "reread192")) { k_calib_reread<<<8192, 256>>>(a, (192ull << 20) / 16, out); bytes += 192ull << 20; }
            else if (!strcmp(pat, "gather1")) { k_calib_gather<1><<<4096, 256>>>(t, 1580800 / 400, 4u << 20, 77u + rep, out); bytes += (4u << 20) * 400.0; }
            else if (!strcmp(pat, "gather7")) { k_calib_gather<7><<<4096, 256>>>(t, 7108000 / 400, 4u << 20, 77u + rep, out); bytes += (4u << 20) * 400.0; }
            else if (!strcmp(pat, "gather109")) { k_calib_gather<109><<<4096, 256>>>(t, 109414400 / 400, 4u << 20, 77u + rep, out); bytes += (4u << 20) * 400.0; }
            else if (!strcmp(pat, "gather1024")) { k_calib_gather<1024><<<4096, 256>>>(t, (uint32_t)((GiB - 400) / 400), 4u << 20, 77u + rep, out); bytes += (4u << 20) * 400.0; }
            else if (!strcmp(pat, "idle")) { usleep(1000); }
            else { fprintf(stderr, "unknown pattern %s\n", pat); return 1; }
        }
        CK(hipDeviceSynchronize());
    }
    const double s = elapsed();
    printf("{\"pattern\": \"%s\", \"seconds\": %.3f, \"launches\": %ld, \"known_GBs\": %.1f}\n", pat, s, launches, bytes / s / 1e9);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 4 && !strcmp(argv[1], "loop")) return loop_mode(argv[2], atof(argv[3]));
    const size_t GiB = 1ull << 30;
    float4 *a, *b;
    float *out;
    CK(hipMalloc(&a, GiB));
    CK(hipMalloc(&b, GiB));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, GiB));
    CK(hipMemset(b, 0, GiB));
    CK(hipDeviceSynchronize());
    const size_t n = GiB / 16;
    for (int rep = 0; rep < 3; ++rep) k_calib_read<<<8192, 256>>>(a, n, out);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) k_calib_write<<<8192, 256>>>(b, n, 1.f + rep);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) k_calib_copy<<<8192, 256>>>(a, b, n);
    CK(hipDeviceSynchronize());
    // the 1 GiB copy has just pushed everything else out of the Infinity Cache: launch 1 of the re-read comes from HBM
    for (int rep = 0; rep < 8; ++rep) k_calib_reread<<<8192, 256>>>(a, (64ull << 20) / 16, out);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) k_calib_l2read<<<8192, 256>>>(a + (128ull << 20) / 16, (2ull << 20) / 16, 32, out);
    CK(hipDeviceSynchronize());
    const float *t = reinterpret_cast<const float *>(b);
    gather<1>(t, 1580800, out);             // ml1m: 3952 x 100 x 4
    gather<7>(t, 7108000, out);             // Netflix shape: 17770 x 100 x 4
    gather<109>(t, 109414400, out);         // Yahoo shape: 136768 x 200 x 4
    gather<1024>(t, GiB - 400, out);
    printf("read / write / copy: 1073741824 bytes per launch and direction, 3 launches each; reread: 67108864 bytes x 8 launches; "
           "l2read: 2097152 bytes x 32 passes x 3 launches\n");
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(out));
    return 0;
}
