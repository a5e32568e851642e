// micro-benchmark: what does a kernel that WAITS on another stream cost the kernels of a busy stream?  (The speculative U step
// of pcr_solver.hip holds its streams behind one-wave gate kernels while the V step runs on the solver's stream.)
// Stream A runs N dependent tiny kernels back to back; meanwhile stream(s) B hold
//   0: nothing   1: a spinning one-wave kernel (relaxed sc1 polls)   2: the same + a kernel queued behind it
//   3: a spinning kernel that polls with agent-scope ACQUIRE loads     4: hipStreamWaitEvent on an event not yet recorded... (n/a: recorded late)
// usage: gate_probe [nB streams = 4] [N = 400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_tiny(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
template <bool ACQ>
__global__ void k_wait(const int* flag) {
    if (threadIdx.x) return;
    const long long t0 = wall_clock64();
    while ((ACQ ? __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
        if (wall_clock64() - t0 > 300000000ll) return;      // 3 s
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ void k_empty() {}
__global__ void k_spin_us(int us) { const long long t0 = wall_clock64(); for (int i = 0; i < (1 << 22) && wall_clock64() - t0 < 100ll * us; ++i) __builtin_amdgcn_s_sleep(16); }
__global__ void k_set(int* flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main(int argc, char** argv) {
    const int nB = argc > 1 ? atoi(argv[1]) : 4, N = argc > 2 ? atoi(argv[2]) : 400;
    hipStream_t A; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    std::vector<hipStream_t> B(nB);
    for (auto& b : B) CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipStream_t H; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&H, hipStreamNonBlocking, hi));
    float* buf; int* flag; const int n = 1 << 24;
    CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4)); CK(hipMalloc(&flag, 8)); hipStream_t H2; CK(hipStreamCreateWithFlags(&H2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipEvent_t eb[8]; for (auto& e : eb) CK(hipEventCreate(&e));
    for (int mode = 0; mode <= 9; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(flag, 0, 8)); CK(hipDeviceSynchronize());
            std::vector<hipStream_t> W = B;
            if (mode == 5) { W.clear(); W.push_back(H); }                 // one waiter, on a high-priority stream
            auto waiters = [&]() {
                int j = 0;
                for (auto b : W) {
                    if (mode == 3) hipLaunchKernelGGL(k_wait<true>, dim3(1), dim3(64), 0, b, flag); else hipLaunchKernelGGL(k_wait<false>, dim3(1), dim3(64), 0, b, flag);
                    if (mode == 2 || mode >= 4) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, b, buf, 64 * 256);
                    if (mode == 4) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, b, buf, 64 * 256);
                    if (mode == 7) CK(hipEventRecord(eb[j++ & 7], b));
                }
            };
            if (mode >= 1 && mode <= 5) waiters();
            if (mode == 9) {                                              // 9: A's work queued BEHIND an event of its own first (the host far ahead)
                hipLaunchKernelGGL(k_wait<false>, dim3(1), dim3(64), 0, A, flag + 1);
            }
            CK(hipEventRecord(e0, A));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(n / 256), dim3(256), 0, A, buf, n);
            if (mode >= 8) {                                              // 8, 9: the B streams (and H) blocked on an EVENT at A's tail, a kernel behind it
                CK(hipEventRecord(eb[0], A));
                std::vector<hipStream_t> X = B; X.push_back(H);
                for (auto b : X) { CK(hipStreamWaitEvent(b, eb[0], 0)); hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, b, buf, 64 * 256); }
                if (mode == 9) hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, H2, flag + 1, 1);
            } else
            if (mode >= 6) waiters();                                     // 6, 7: the waiters arrive while A has a backlog (7: + an event behind each)
            hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, A, flag, 1);
            if (mode == 7) for (size_t j = 0; j < W.size(); ++j) CK(hipStreamWaitEvent(A, eb[j & 7], 0));
            CK(hipEventRecord(e1, A)); CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("mode %d: %8.2f us per kernel on the busy stream\n", mode, 1e3 * ms / N);
        }
    }
    // mode 10 = the probe pcr_solver.hip's pick_lanes runs (shares_pipe): a chain of 24 EMPTY 16384-workgroup kernels on A, alone and
    // with a spinning one-wave kernel (nothing queued behind it) on one other stream -- every B stream and H in turn
    {
        std::vector<hipStream_t> X = B; X.push_back(H); X.push_back(H2);
        for (int wgs : {16384, 65536}) {
        auto chain = [&]() { CK(hipEventRecord(e0, A)); for (int i = 0; i < 24; ++i) hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(64), 0, A); CK(hipEventRecord(e1, A));
                             CK(hipEventSynchronize(e1)); float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / 24; };
        chain();
        const double base = chain();
        for (size_t j = 0; j < X.size(); ++j) {
            hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, X[j], 2400);
            const double with = chain();
            CK(hipStreamSynchronize(X[j]));
            printf("mode 10 (%d workgroups): stream %zu%s: %.1f us per empty kernel alone, %.1f with that queue held\n", wgs, j, j == B.size() ? " (high priority)" : "", base, with);
        }
        }
    }
    return 0;
}
