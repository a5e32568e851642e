// pcr_initial.cpp -- pcr_initial / pcr_initial_rows: the reference's initial() stream (util.cpp:80-93), bit for bit, by several threads.
// A translation unit of its own because it is compiled by g++ (the rest of the host data path by hipcc's clang, whose parser code is
// the faster one): libstdc++'s generate_canonical evaluates two long-double logarithms of constants on EVERY call; g++ folds them,
// clang calls logl -- 3.5 x on this stream (10 against 36 ms per million values).  No GPU calls in this file.
#include "pcr_host.h"

#include <algorithm>
#include <cstdint>
#include <functional>
#include <limits>
#include <random>
#include <vector>

namespace {
template <class F>
int guarded(const char* what, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { pcr_set_error(std::string(what) + ": out of memory"); } catch (...) {} return PCR_ERR_NOMEM; }
    catch (const std::exception& e) { try { pcr_set_error(std::string(what) + ": " + e.what()); } catch (...) {} return PCR_ERR_ARG; }
    catch (...) { return PCR_ERR_ARG; }
}
}  // namespace

// util.cpp:80-93: every call constructs a default-seeded std::default_random_engine (= minstd_rand0, seed 1) and draws n * k values
// from std::normal_distribution<double>(0, 1).  Init parity depends on libstdc++'s generate_canonical / polar method, so the std
// facilities are called directly -- but by several threads: the stream is sequential only in WHERE an output lands.
//   * libstdc++'s normal_distribution draws points in the unit square until one falls inside the circle, and every try takes exactly
//     FOUR engine draws (two generate_canonical<double, 53> of two 31-bit draws each): try number a starts at engine draw 4a
//     whatever happened before, and minstd_rand0 is x -> 16807 x mod (2^31 - 1): the state after d draws is 16807^d mod (2^31 - 1).
//   * an accepted try yields two outputs (y * m, then the saved x * m), a rejected one none.
// So the tries are cut into ranges; a thread seeds an engine at its range's first draw, runs the SAME std::normal_distribution
// over it and counts the engine draws (a counting URBG around the engine); pass 1 counts each range's accepted tries, pass 2
// writes every range's outputs at its prefix offset.  A range's last call may run past its end (rejected tries at the start of
// the next range, then that range's first accepted try): that pair belongs to the next range, which produces it too, and is dropped.
namespace {
struct CountingMinstd {                                    // a URBG: minstd_rand0 + the number of draws taken
    typedef std::minstd_rand0::result_type result_type;
    std::minstd_rand0 eng;
    uint64_t draws = 0;
    explicit CountingMinstd(result_type state) : eng(state) {}
    static constexpr result_type min() { return std::minstd_rand0::min(); }
    static constexpr result_type max() { return std::minstd_rand0::max(); }
    result_type operator()() { ++draws; return eng(); }
};
// state of a default-seeded minstd_rand0 after d draws
uint32_t minstd_state_after(uint64_t d) {
    const uint64_t m = 2147483647ull;
    uint64_t r = 1, a = 16807;
    for (; d; d >>= 1) { if (d & 1) r = r * a % m; a = a * a % m; }
    return (uint32_t)r;                                   // x0 = the default seed 1
}
// the outputs of tries [a0, a1): fn(i, value) for output i = 0, 1, ... of the range; returns how many outputs the range owns
template <class F>
int64_t normal_range(uint64_t a0, uint64_t a1, F&& fn) {
    CountingMinstd g(minstd_state_after(4 * a0));
    std::normal_distribution<double> dist(0.0, 1.0);
    const uint64_t quota = 4 * (a1 - a0);
    int64_t out = 0;
    while (g.draws < quota) {
        const double y = dist(g);                          // an accepted try: two outputs (the second comes from the saved value, no draw)
        if (g.draws > quota) break;                        // ... that began past the end of this range: the next range's first pair
        const double x = dist(g);
        fn(out, y); fn(out + 1, x);
        out += 2;
    }
    return out;
}
// X[i - first] = output i of the reference's stream for i in [first, first + count); total outputs of the stream wanted: first + count
void normal_stream(double* X, int64_t first, int64_t count) {
    const int64_t need = first + count;
    int T = pcr_host_threads();
    if (need < ((int64_t)1 << 22)) T = 1;                 // (up to ~50 ms of serial work: threads would not pay)
    if (T == 1) {                                          // the reference's own loop
        std::default_random_engine generator;
        std::normal_distribution<double> distribution(0.0, 1.0);
        for (int64_t i = 0; i < need; ++i) { const double v = distribution(generator); if (i >= first) X[i - first] = v; }
        return;
    }
    // tries that yield `need` outputs: acceptance is pi / 4; 1 % + 4096 on top, and a serial tail if that still falls short
    const uint64_t tries = (uint64_t)((double)need / 2.0 / 0.7853981633974483 * 1.01) + 4096;
    const int R = T * 4;
    std::vector<uint64_t> cut((size_t)R + 1);
    for (int r = 0; r <= R; ++r) cut[(size_t)r] = tries * (uint64_t)r / (uint64_t)R;
    std::vector<int64_t> ofs((size_t)R + 1, 0);
    auto run = [&](const std::function<void(int)>& body) { pcr_parallel_tasks(R, T, body); };
    // pass 1 needs only which tries are accepted: the distribution's own test (x * x + y * y inside the unit circle, not the origin) on
    // the same two generate_canonical values, without the sqrt / log of an accepted try
    run([&](int r) {
        CountingMinstd g(minstd_state_after(4 * cut[(size_t)r]));
        int64_t acc = 0;
        for (uint64_t a = cut[(size_t)r]; a < cut[(size_t)r + 1]; ++a) {
            const double x = 2.0 * std::generate_canonical<double, std::numeric_limits<double>::digits>(g) - 1.0;
            const double y = 2.0 * std::generate_canonical<double, std::numeric_limits<double>::digits>(g) - 1.0;
            const double r2 = x * x + y * y;
            acc += !(r2 > 1.0 || r2 == 0.0);
        }
        ofs[(size_t)r + 1] = 2 * acc;
    });
    for (int r = 0; r < R; ++r) ofs[(size_t)r + 1] += ofs[(size_t)r];
    run([&](int r) {
        const int64_t base = ofs[(size_t)r];
        if (base >= need || ofs[(size_t)r + 1] <= first) return;
        normal_range(cut[(size_t)r], cut[(size_t)r + 1], [&](int64_t i, double v) { const int64_t g = base + i; if (g >= first && g < need) X[g - first] = v; });
    });
    // (not seen in practice -- the slack is ~40 standard deviations of the accepted count -- but the stream simply goes on)
    uint64_t a = tries;
    for (int64_t have = ofs[(size_t)R]; have < need;) {
        const uint64_t a2 = a + (uint64_t)(need - have) + 4096;
        have += normal_range(a, a2, [&](int64_t j, double v) { const int64_t g = have + j; if (g >= first && g < need) X[g - first] = v; });
        a = a2;
    }
}
}  // namespace

extern "C" int pcr_initial(double* X, int64_t n, int64_t k) {
    if (!X || n < 0 || k < 0) { pcr_set_error("pcr_initial: bad argument"); return PCR_ERR_ARG; }
    return guarded("pcr_initial", [&]() -> int { normal_stream(X, 0, n * k); return PCR_OK; });
}

extern "C" int pcr_initial_rows(double* X, int64_t n, int64_t k, int64_t row0, int64_t nrows) {
    if (!X || n < 0 || k < 0 || row0 < 0 || nrows < 0 || row0 + nrows > n) { pcr_set_error("pcr_initial_rows: bad argument"); return PCR_ERR_ARG; }
    return guarded("pcr_initial_rows", [&]() -> int { normal_stream(X, row0 * k, nrows * k); return PCR_OK; });
}

