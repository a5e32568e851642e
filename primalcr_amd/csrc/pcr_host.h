// pcr_host.h -- host-side data structures shared by the loader, the solver and the CLIs.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "primalcr.h"

// std::vector whose resize(n) leaves the new elements UNINITIALISED: the loader's nnz-sized arrays are written exactly once, by the
// threads that parse / scatter into them (a zero-fill of 8.4 GB on one thread is seconds at the Yahoo!Music shape, and it
// would first-touch every page on the calling thread's NUMA node).
template <class T>
struct PcrNoInit : std::allocator<T> {
    template <class U> struct rebind { using other = PcrNoInit<U>; };
    PcrNoInit() = default;
    template <class U> PcrNoInit(const PcrNoInit<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using pcr_vec = std::vector<T, PcrNoInit<T>>;

// user-major CSR in the reference's SparseMat layout (util.h:390-413); `item` is
// SparseMat::rows, the user id per rating (SparseMat::cols) is implied by `index`.
struct PcrCsr {
    int64_t d1 = 0, d2 = 0;
    std::vector<int64_t> index;   // d1 + 1
    pcr_vec<int32_t> item;        // nnz
    pcr_vec<double> val;          // nnz
    int64_t nnz() const { return index.empty() ? 0 : index.back(); }
};

struct pcr_dataset {
    PcrCsr train, test;
    int64_t tnnz_file = 0;        // test entries read from the file (T.nnz, pcrpp.cpp:865)
};

// dense rank of the rating bucket inside each user: lround(val) for PrimalCR++
// (pcrpp.cpp:38-49,182-189), the raw double for PrimalCR (pcr.cpp:23 compares doubles).
// level[z] in [0, T_u); run_ofs[u]..run_ofs[u+1] indexes run_start, which holds T_u + 1
// cumulative per-level counts for user u (run_start[run_ofs[u] + T_u] == len_u).
struct PcrLevels {
    std::vector<uint16_t> level;      // nnz
    std::vector<int64_t> run_ofs;     // d1 + 1
    std::vector<int32_t> run_start;   // sum_u (T_u + 1)
    std::vector<double> lev_val;      // same slots as run_start: the key of level l of user u at run_ofs[u] + l (slot T_u unused)
    int max_levels = 0;
    bool integer_valued = true;       // every rating of the range equals its own lround(): raw levels = rounded levels
};
int pcr_build_levels(const PcrCsr& X, int64_t u0, int64_t u1, int solver_type, PcrLevels& out, std::string& err);

// host set-up helpers: fn(thread, lo, hi) over contiguous pieces of [0, n); the thread count set-up work uses
void pcr_parallel_ranges(int64_t n, int nthreads, const std::function<void(int, int64_t, int64_t)>& fn);
void pcr_parallel_tasks(int ntasks, int nthreads, const std::function<void(int)>& fn);
int pcr_host_threads();

void pcr_set_error(const std::string& msg);

// launch knobs set through pcr_tune() (include/primalcr.h): consulted by the solver when it is created
bool pcr_tune_get(const char* key, std::string* out);
int pcr_tune_int(const char* key, int dflt);
