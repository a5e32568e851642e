// Dev tool (GPU box): what does a fork / join between two HIP streams cost per CG iteration?  (run the even user
// tiles of one Hessian-vector product -- SDDMM, sweep, SpMM -- on the solver's stream and the odd ones on a side stream, joined
// before k_spmm_fin; NOTES.md round 1 saw +0.2 ms for a host-side fork.)
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_build/join_probe tools/ubench/join_probe.hip
// One-workgroup kernels that spin for a given time stand in for the five kernels of a CG iteration (25.7 / 13.5 / 26.5 / 6.3 / 6.3 us
// on the ml1m shape, profiles/r05_a_ml1m_f32_kernel_stats.csv) -- no bandwidth is shared, so the figures are the cost of the stream
// mechanics alone:
//   one stream          the five kernels back to back, 10 iterations queued, as the solver does today
//   two streams         each of the three tile-parallel kernels at HALF its duration on either stream; per iteration one device-side
//                       fork (side stream waits for an event of the solver's stream) and one join (the solver's stream waits for
//                       the side stream's event) before the two short kernels; ideal = half the tile-parallel time + the short ones
//   two streams, host   the same with the join done by the host (hipEventSynchronize), as for_bins joins its lanes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_busy(long long ticks) {
    const long long t0 = wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

int main() {
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    auto ticks = [&](double us) { return (long long)(us * khz / 1e3); };
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork_ev[16], join_ev[16];
    for (int i = 0; i < 16; ++i) { CK(hipEventCreateWithFlags(&fork_ev[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming)); }
    const double par[3] = {25.7, 13.5, 26.5}, ser[2] = {6.3, 6.3};
    const int IT = 10, REP = 30;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto us_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(now() - t).count(); };
    // warm up both streams
    for (int i = 0; i < 20; ++i) { k_busy<<<1, 64, 0, s0>>>(ticks(5)); k_busy<<<1, 64, 0, s1>>>(ticks(5)); }
    CK(hipDeviceSynchronize());
    double one = 0, two = 0, two_host = 0;
    for (int rep = 0; rep < REP; ++rep) {
        auto t = now();
        for (int it = 0; it < IT; ++it) {
            for (double d : par) k_busy<<<1, 64, 0, s0>>>(ticks(d));
            for (double d : ser) k_busy<<<1, 64, 0, s0>>>(ticks(d));
        }
        CK(hipStreamSynchronize(s0));
        one += us_since(t);
        t = now();
        for (int it = 0; it < IT; ++it) {
            CK(hipEventRecord(fork_ev[it], s0));                               // everything before this iteration (the CG update) is done
            CK(hipStreamWaitEvent(s1, fork_ev[it], 0));
            for (double d : par) { k_busy<<<1, 64, 0, s0>>>(ticks(d / 2)); k_busy<<<1, 64, 0, s1>>>(ticks(d / 2)); }
            CK(hipEventRecord(join_ev[it], s1));
            CK(hipStreamWaitEvent(s0, join_ev[it], 0));
            for (double d : ser) k_busy<<<1, 64, 0, s0>>>(ticks(d));
        }
        CK(hipStreamSynchronize(s0));
        two += us_since(t);
        t = now();
        for (int it = 0; it < IT; ++it) {
            CK(hipEventRecord(fork_ev[it], s0));
            CK(hipStreamWaitEvent(s1, fork_ev[it], 0));
            for (double d : par) { k_busy<<<1, 64, 0, s0>>>(ticks(d / 2)); k_busy<<<1, 64, 0, s1>>>(ticks(d / 2)); }
            CK(hipEventRecord(join_ev[it], s1));
            CK(hipEventSynchronize(join_ev[it]));
            for (double d : ser) k_busy<<<1, 64, 0, s0>>>(ticks(d));
        }
        CK(hipStreamSynchronize(s0));
        two_host += us_since(t);
    }
    const double sum_par = par[0] + par[1] + par[2], sum_ser = ser[0] + ser[1];
    printf("kernel time per CG iteration: %.1f us on one stream, %.1f us ideal on two (tile-parallel kernels halved)\n", sum_par + sum_ser, sum_par / 2 + sum_ser);
    printf("one stream                      : %7.1f us per iteration (+%.1f over its kernels)\n", one / REP / IT, one / REP / IT - sum_par - sum_ser);
    printf("two streams, device-side join   : %7.1f us per iteration (+%.1f over the ideal: one fork + one join)\n", two / REP / IT, two / REP / IT - sum_par / 2 - sum_ser);
    printf("two streams, host-side join     : %7.1f us per iteration (+%.1f over the ideal)\n", two_host / REP / IT, two_host / REP / IT - sum_par / 2 - sum_ser);
    return 0;
}
