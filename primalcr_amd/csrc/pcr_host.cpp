// pcr_host.cpp -- host data path of libprimalcr: meta/ratings loader, CSR conversion, level
// ranking, model file I/O, initial(), user partitioning.  No GPU calls in this file.
//
// Mirrors the *formats and semantics* of the reference (file:line cited per function); the
// reference's vector<vector<double>> / smat_t containers are not reproduced.
#include "pcr_host.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <random>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static thread_local std::string g_err;
void pcr_set_error(const std::string& msg) { g_err = msg; }

extern "C" const char* pcr_last_error(void) { return g_err.c_str(); }

// No C++ exception may cross the C ABI: a header that promises 2^60 ratings, a meta file with a negative count or a machine
// that is simply out of memory must come back as an error code, not as std::terminate (found by the malformed-input test,
// tests/test_host_abi.py, which also runs under AddressSanitizer).
template <class F>
static int guarded(const char* what, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { pcr_set_error(std::string(what) + ": out of memory"); } catch (...) {} return PCR_ERR_NOMEM; }
    catch (const std::exception& e) { try { pcr_set_error(std::string(what) + ": " + e.what()); } catch (...) {} return PCR_ERR_ARG; }
    catch (...) { return PCR_ERR_ARG; }
}
// Worker threads of the host data path: `n_workers` std::threads run body(worker index); an exception thrown on a worker (a failed
// allocation inside a counting sort, say) must not end the process through std::terminate -- the first one is carried back and
// rethrown on the calling thread, where the entry point's guard turns it into an error code.
template <class Body>
static void run_workers(int n_workers, Body&& body) {
    if (n_workers <= 1) { body(0); return; }
    std::exception_ptr first;
    std::mutex mu;
    auto safe = [&](int w) {
        try { body(w); }
        catch (...) { std::lock_guard<std::mutex> lk(mu); if (!first) first = std::current_exception(); }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)n_workers - 1);
    try {
        for (int w = 1; w < n_workers; ++w) th.emplace_back(safe, w);
    } catch (...) {                                        // (thread creation failed: finish with the threads that exist)
        std::lock_guard<std::mutex> lk(mu);
        if (!first) first = std::current_exception();
    }
    safe(0);
    for (auto& x : th) x.join();
    if (first) std::rethrow_exception(first);
}

static const int64_t kMaxDim = ((int64_t)1 << 31) - 2;             // ids are int32 throughout (the device solver's limit too)
extern "C" const char* pcr_version(void) { return "primalcr-mi355x 0.1 (gfx950)"; }

// pmf.h:27-48
extern "C" void pcr_params_default(pcr_params* p) {
    p->solver_type = PCR_SOLVER_PCRPP;
    p->k = 10;
    p->threads = 4;
    p->maxiter = 10;
    p->lambda = 5000;
    p->do_predict = 1;
    p->verbose = 0;
    p->stepsize = 1.0;
    p->ndcg_k = 10;
    p->precision = PCR_F32;
    p->device = 0;
    p->cg_max_iter = 10;
    p->cg_tol = 0.01;
}

// (initial(): pcr_initial.cpp -- a translation unit of its own, compiled by g++)

// ------------------------------------------------------------------------------------------
// CSR conversion
// ------------------------------------------------------------------------------------------

// fn(t, lo, hi) over T contiguous pieces of [0, n), one std::thread each (T = 1: on the calling thread)
template <class F>
static void run_pieces(int64_t n, int T, F&& fn) {
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n));
    if (T == 1) { fn(0, (int64_t)0, n); return; }
    run_workers(T, [&](int t) { fn(t, n * t / T, n * (t + 1) / T); });
}
static int pieces_for(int64_t n, int threads) { return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(threads, 64), n / 65536 + 1)); }

// index[u] = number of entries whose key is below u, for a NON-DECREASING key sequence key[0, n) with values in [0, d1):
// every piece writes the index entries of the keys that first appear in it (each entry is written exactly once).
static void index_of_sorted_keys(const int32_t* key, int64_t n, int64_t d1, int64_t* index, int T) {
    if (n == 0) { std::fill(index, index + d1 + 1, (int64_t)0); return; }
    run_pieces(n, T, [&](int, int64_t lo, int64_t hi) {
        int64_t prev = lo == 0 ? -1 : key[lo - 1];
        for (int64_t z = lo; z < hi; ++z) {
            const int64_t k = key[z];
            if (k != prev) { for (int64_t u = prev + 1; u <= k; ++u) index[u] = z; prev = k; }
        }
        if (hi == n) for (int64_t u = prev + 1; u <= d1; ++u) index[u] = n;
    });
}

// util.h:223-247 sorts entries by (row, col); util.cpp:229-243 walks them user by user: items ascending inside a user.
// `threads` host threads throughout:
//   * the usual case -- the file is already ordered by (user, item), as the reference's own data sets and the generators'
//     are -- is detected piece by piece during the range check and needs no data movement at all: the parsed item / value
//     arrays ARE the CSR (they are adopted when the caller hands them over in X, copied in parallel otherwise) and the row
//     pointers come from the first occurrence of every user;
//   * anything else: entries are scattered into user-range buckets (per piece and bucket counts -> offsets, file order kept
//     inside a bucket), then every bucket -- on its own thread -- does its counting sort by user and, where a user's items are
//     not ascending, a stable sort by item (equal (user, item) pairs keep their file order).
// adopt = item / val point at X.item / X.val (sized nnz) already.
static int build_train_csr(int64_t d1, int64_t d2, int64_t nnz, const int32_t* user, const int32_t* item,
                           const double* val, PcrCsr& X, int threads, bool adopt) {
    X.d1 = d1; X.d2 = d2;
    X.index.resize((size_t)d1 + 1);
    const int T = pieces_for(nnz, threads);
    struct Piece { int64_t bad = -1; bool sorted = true; };
    std::vector<Piece> pc((size_t)T);
    run_pieces(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        Piece& P = pc[(size_t)t];
        int64_t pu = lo > 0 ? user[lo - 1] : -1, pi = lo > 0 ? item[lo - 1] : -1;      // (the pair before the piece: the seam counts too)
        for (int64_t z = lo; z < hi; ++z) {
            const int64_t u = user[z], i = item[z];
            if (u < 0 || u >= d1 || i < 0 || i >= d2) { P.bad = z; return; }
            if (u < pu || (u == pu && i < pi)) P.sorted = false;
            pu = u; pi = i;
        }
    });
    bool sorted = true;
    for (const Piece& P : pc) {
        if (P.bad >= 0) {
            int64_t first_bad = P.bad;
            for (const Piece& Q : pc) if (Q.bad >= 0) first_bad = std::min(first_bad, Q.bad);
            pcr_set_error("rating " + std::to_string(first_bad) + " has user/item id outside the meta dimensions");
            return PCR_ERR_ARG;
        }
        sorted = sorted && P.sorted;
    }
    if (sorted) {
        index_of_sorted_keys(user, nnz, d1, X.index.data(), T);
        if (!adopt) {
            X.item.resize((size_t)nnz); X.val.resize((size_t)nnz);
            run_pieces(nnz, T, [&](int, int64_t lo, int64_t hi) {
                std::copy(item + lo, item + hi, X.item.data() + lo);
                std::copy(val + lo, val + hi, X.val.data() + lo);
            });
        }
        return PCR_OK;
    }
    // ---- general path
    const int64_t NB = std::max<int64_t>(1, std::min<int64_t>(d1, (int64_t)T * 8));   // user-range buckets (more than threads: balance)
    const int64_t bw = (d1 + NB - 1) / NB;
    std::vector<int64_t> cnt((size_t)T * NB, 0);
    run_pieces(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t* c = cnt.data() + (size_t)t * NB;
        for (int64_t z = lo; z < hi; ++z) c[user[z] / bw]++;
    });
    std::vector<int64_t> bstart((size_t)NB + 1, 0);
    for (int64_t b = 0; b < NB; ++b) {
        int64_t acc = bstart[b];
        for (int t = 0; t < T; ++t) { const int64_t c = cnt[(size_t)t * NB + b]; cnt[(size_t)t * NB + b] = acc; acc += c; }
        bstart[b + 1] = acc;
    }
    pcr_vec<int32_t> bu((size_t)nnz), bi((size_t)nnz);
    pcr_vec<double> bv((size_t)nnz);
    run_pieces(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t* c = cnt.data() + (size_t)t * NB;
        for (int64_t z = lo; z < hi; ++z) {
            const int64_t q = c[user[z] / bw]++;
            bu[q] = user[z]; bi[q] = item[z]; bv[q] = val[z];
        }
    });
    pcr_vec<int32_t> oi((size_t)nnz);
    pcr_vec<double> ov((size_t)nnz);
    std::atomic<int64_t> next{0};
    auto bucket_work = [&]() {
        std::vector<int64_t> cur;
        std::vector<std::pair<int32_t, double>> tmp;
        for (;;) {
            const int64_t b = next.fetch_add(1);
            if (b >= NB) break;
            const int64_t ulo = b * bw, uhi = std::min(d1, ulo + bw), a = bstart[b], e = bstart[b + 1];
            if (ulo >= uhi) continue;
            cur.assign((size_t)(uhi - ulo) + 1, 0);
            for (int64_t z = a; z < e; ++z) cur[(size_t)(bu[z] - ulo) + 1]++;
            int64_t acc = a;                                             // entries of users below ulo = the bucket's start
            for (int64_t u = ulo; u < uhi; ++u) { const int64_t c = cur[(size_t)(u - ulo) + 1]; X.index[u] = acc; cur[(size_t)(u - ulo)] = acc; acc += c; }
            for (int64_t z = a; z < e; ++z) { const int64_t q = cur[(size_t)(bu[z] - ulo)]++; oi[q] = bi[z]; ov[q] = bv[z]; }
            for (int64_t u = ulo; u < uhi; ++u) {
                const int64_t ua = X.index[u], ub = u + 1 < uhi ? X.index[u + 1] : e;
                bool asc = true;
                for (int64_t q = ua + 1; q < ub && asc; ++q) asc = oi[q - 1] <= oi[q];
                if (asc) continue;
                tmp.resize((size_t)(ub - ua));
                for (int64_t q = ua; q < ub; ++q) tmp[(size_t)(q - ua)] = {oi[q], ov[q]};
                std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
                for (int64_t q = ua; q < ub; ++q) { oi[q] = tmp[(size_t)(q - ua)].first; ov[q] = tmp[(size_t)(q - ua)].second; }
            }
        }
    };
    run_workers(T, [&](int) { bucket_work(); });
    for (int64_t u = NB * bw; u < d1; ++u) X.index[u] = nnz;              // (none: NB * bw >= d1)
    X.index[d1] = nnz;
    X.item.swap(oi); X.val.swap(ov);
    return PCR_OK;
}

// util.cpp:250-274: the test file is assumed user-sorted; the reference's cursor walks the users 0, 1, ... and appends entries
// while their user id is not above the cursor -- so an entry lands in the row of the LARGEST user id seen so far (itself
// included), and the scan ends for good at the first id that is not a user (>= d1): that entry and everything behind it are
// dropped.  Mirrored, in parallel: per-piece maxima -> carried-in running maximum -> row of every entry -> row pointers as for
// sorted keys.  adopt = item / val point at X.item / X.val (sized nnz); they are cut at the end of the scan.
static void build_test_csr(int64_t d1, int64_t d2, int64_t nnz, const int32_t* user, const int32_t* item,
                           const double* val, PcrCsr& X, int threads, bool adopt) {
    X.d1 = d1; X.d2 = d2;
    X.index.resize((size_t)d1 + 1);
    const int T = pieces_for(nnz, threads);
    std::vector<int64_t> pmax((size_t)T + 1, -1);
    run_pieces(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t m = -1;
        for (int64_t z = lo; z < hi; ++z) m = std::max<int64_t>(m, user[z]);
        pmax[(size_t)t + 1] = m;
    });
    pmax[0] = 0;                                                          // the cursor starts at user 0
    for (int t = 0; t < T; ++t) pmax[(size_t)t + 1] = std::max(pmax[(size_t)t], pmax[(size_t)t + 1]);
    pcr_vec<int32_t> row((size_t)nnz);
    std::vector<int64_t> stop((size_t)T, nnz);
    run_pieces(nnz, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t m = pmax[(size_t)t];
        for (int64_t z = lo; z < hi; ++z) {
            m = std::max<int64_t>(m, user[z]);
            if (m >= d1) { stop[(size_t)t] = z; for (; z < hi; ++z) row[z] = 0; return; }
            row[z] = (int32_t)m;
        }
    });
    int64_t keep = nnz;
    for (int t = 0; t < T; ++t) keep = std::min(keep, stop[(size_t)t]);
    index_of_sorted_keys(row.data(), keep, d1, X.index.data(), T);
    if (!adopt) {
        X.item.resize((size_t)keep); X.val.resize((size_t)keep);
        run_pieces(keep, T, [&](int, int64_t lo, int64_t hi) {
            std::copy(item + lo, item + hi, X.item.data() + lo);
            std::copy(val + lo, val + hi, X.val.data() + lo);
        });
    } else {
        X.item.resize((size_t)keep); X.val.resize((size_t)keep);
    }
}

static int check_test_ids(int64_t d2, int64_t tnnz, const int32_t* tuser, const int32_t* titem, int threads) {
    const int T = pieces_for(tnnz, threads);
    std::vector<int64_t> bad((size_t)T, -1);
    run_pieces(tnnz, T, [&](int t, int64_t lo, int64_t hi) {
        for (int64_t z = lo; z < hi; ++z)
            if (titem[z] < 0 || titem[z] >= d2 || tuser[z] < 0) { bad[(size_t)t] = z; return; }
    });
    for (int64_t b : bad)
        if (b >= 0) {
            pcr_set_error("test rating " + std::to_string(b) + " has user/item id outside the meta dimensions");
            return PCR_ERR_ARG;
        }
    return PCR_OK;
}

extern "C" int pcr_dataset_from_triplets(int64_t d1, int64_t d2, int64_t nnz, const int32_t* user,
                                         const int32_t* item, const double* val, int64_t tnnz,
                                         const int32_t* tuser, const int32_t* titem, const double* tval,
                                         pcr_dataset** out) {
    if (!out || d1 < 0 || d2 < 0 || nnz < 0 || tnnz < 0 || (nnz > 0 && (!user || !item || !val)) ||
        (tnnz > 0 && (!tuser || !titem || !tval))) {
        pcr_set_error("pcr_dataset_from_triplets: bad argument");
        return PCR_ERR_ARG;
    }
    if (d1 > kMaxDim || d2 > kMaxDim) { pcr_set_error("pcr_dataset_from_triplets: more than 2^31 - 2 users or items"); return PCR_ERR_UNSUPPORTED; }
    return guarded("pcr_dataset_from_triplets", [&]() -> int {
        std::unique_ptr<pcr_dataset> ds(new pcr_dataset());
        const int threads = pcr_host_threads();
        int rc = build_train_csr(d1, d2, nnz, user, item, val, ds->train, threads, false);
        if (rc != PCR_OK) return rc;
        rc = check_test_ids(d2, tnnz, tuser, titem, threads);
        if (rc != PCR_OK) return rc;
        build_test_csr(d1, d2, tnnz, tuser, titem, tval, ds->test, threads, false);
        ds->tnnz_file = tnnz;
        *out = ds.release();
        return PCR_OK;
    });
}

// The same data set from arrays that already ARE the reference's SparseMat layout (util.h:390-413: what convert() leaves,
// util.cpp:219-274): index[d1+1], item[nnz], val[nnz], items ascending inside a user for the training set (checked; the
// test set is taken as given).  No triplet round trip for callers that hold CSRs (100 M-rating synthetic sets).
extern "C" int pcr_dataset_from_csr(int64_t d1, int64_t d2, const int64_t* index, const int32_t* item, const double* val,
                                    const int64_t* tindex, const int32_t* titem, const double* tval, pcr_dataset** out) {
    if (!out || d1 < 0 || d2 < 0 || !index || index[0] != 0 || (tindex && tindex[0] != 0)) {
        pcr_set_error("pcr_dataset_from_csr: bad argument");
        return PCR_ERR_ARG;
    }
    const int64_t nnz = index[d1], tnnz = tindex ? tindex[d1] : 0;
    if (nnz < 0 || tnnz < 0 || (nnz > 0 && (!item || !val)) || (tnnz > 0 && (!titem || !tval))) {
        pcr_set_error("pcr_dataset_from_csr: bad argument");
        return PCR_ERR_ARG;
    }
    for (int64_t u = 0; u < d1; ++u) {
        if (index[u + 1] < index[u] || (tindex && tindex[u + 1] < tindex[u])) { pcr_set_error("pcr_dataset_from_csr: index not monotone"); return PCR_ERR_ARG; }
        for (int64_t z = index[u]; z < index[u + 1]; ++z)
            if (item[z] < 0 || item[z] >= d2 || (z > index[u] && item[z] <= item[z - 1])) {
                pcr_set_error("pcr_dataset_from_csr: user " + std::to_string(u) + ": item ids must be ascending and inside [0, d2)");
                return PCR_ERR_ARG;
            }
    }
    for (int64_t z = 0; z < tnnz; ++z)
        if (titem[z] < 0 || titem[z] >= d2) { pcr_set_error("pcr_dataset_from_csr: test item id outside [0, d2)"); return PCR_ERR_ARG; }
    if (d1 > kMaxDim || d2 > kMaxDim) { pcr_set_error("pcr_dataset_from_csr: more than 2^31 - 2 users or items"); return PCR_ERR_UNSUPPORTED; }
    return guarded("pcr_dataset_from_csr", [&]() -> int {
        std::unique_ptr<pcr_dataset> ds(new pcr_dataset());
        ds->train.d1 = ds->test.d1 = d1; ds->train.d2 = ds->test.d2 = d2;
        ds->train.index.assign(index, index + d1 + 1);
        ds->train.item.assign(item, item + nnz);
        ds->train.val.assign(val, val + nnz);
        if (tindex) ds->test.index.assign(tindex, tindex + d1 + 1); else ds->test.index.assign(d1 + 1, 0);
        ds->test.item.assign(titem, titem + tnnz);
        ds->test.val.assign(tval, tval + tnnz);
        ds->tnnz_file = tnnz;
        *out = ds.release();
        return PCR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// launch knobs (pcr_tune): a key/value table PER CALLING THREAD, which pcr_solver_create (on that thread) snapshots into the
// solver -- no process-global mutable state: two threads that create solvers with different knobs do not see each other's
// ------------------------------------------------------------------------------------------
#include <map>
static std::map<std::string, std::string>& tune_table() { static thread_local std::map<std::string, std::string> t; return t; }
static const char* const TUNE_KEYS[] = {"ustep_mode", "cluster_k", "cluster_users", "ubins", "ustep_gram", "spmm_tiles", "spmm_chunk", "sddmm_csc",
                                        "lanes", "pipeline", "window_cache", "prepare_merged", "resort_window", "allreduce_chunks", "p2p_ll",
                                        "p2p_timeout_ms", "p2p_queue_budget", "count_rows", "debug", "win16", "ustep_win_lds", "plan_key64", "ustep_newton", "vblock_users", "fault_cluster_member", "fault_p2p_skip", "fault_p2p_coarse", nullptr};
extern "C" int pcr_tune(const char* key, const char* value) {
    if (!key) { pcr_set_error("pcr_tune: null key"); return PCR_ERR_ARG; }
    bool known = false;
    for (const char* const* k = TUNE_KEYS; *k; ++k) known = known || !strcmp(*k, key);
    if (!known) { pcr_set_error(std::string("pcr_tune: unknown key '") + key + "'"); return PCR_ERR_ARG; }
    if (value) tune_table()[key] = value; else tune_table().erase(key);
    return PCR_OK;
}
bool pcr_tune_get(const char* key, std::string* out) {
    auto it = tune_table().find(key);
    if (it == tune_table().end()) return false;
    if (out) *out = it->second;
    return true;
}
int pcr_tune_int(const char* key, int dflt) {
    std::string v;
    return pcr_tune_get(key, &v) ? atoi(v.c_str()) : dflt;
}

// ------------------------------------------------------------------------------------------
// text loader (util.cpp:6-25, util.h:118-131, util.h:360-371)
// ------------------------------------------------------------------------------------------

// A rating file as one read-only byte range: mmap for regular files (no copy, no zero-filled staging buffer; the parser
// threads fault their own pieces in), a heap buffer for anything that cannot be mapped (pipes, /proc, empty files).
struct TextFile {
    const char* p = nullptr;
    size_t len = 0;
    void* map = nullptr;
    size_t map_len = 0;
    pcr_vec<char> heap;
    ~TextFile() { if (map) munmap(map, map_len); }
    int open(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) { pcr_set_error("can't open " + path + ": " + strerror(errno)); return PCR_ERR_IO; }
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)sb.st_size, MADV_WILLNEED);
                map = m; map_len = (size_t)sb.st_size; p = static_cast<const char*>(m); len = map_len;
                ::close(fd);
                return PCR_OK;
            }
        }
        size_t cap = (size_t)1 << 20, got = 0;
        heap.resize(cap);
        for (;;) {
            const ssize_t n = ::read(fd, heap.data() + got, cap - got);
            if (n < 0) { if (errno == EINTR) continue; ::close(fd); pcr_set_error("can't read " + path + ": " + strerror(errno)); return PCR_ERR_IO; }
            if (n == 0) break;
            got += (size_t)n;
            if (got == cap) { cap *= 2; heap.resize(cap); }
        }
        ::close(fd);
        p = heap.data(); len = got;
        return PCR_OK;
    }
};

// strtol / strtod need a terminated string and the mapped file has none: the (rare) libc path parses a bounded copy.
struct SlowCopy {
    char buf[512];
    size_t n;
    SlowCopy(const char* p, const char* end) { n = std::min<size_t>(sizeof(buf) - 1, (size_t)(end - p)); memcpy(buf, p, n); buf[n] = 0; }
};

// One rating "user item value" (util.h:126, util.h:367).  Fast path for the common "digits digits
// [-]digits[.digits]" shape, strtol/strtod for anything else (exponents, inf, ...).
static inline bool parse_one(const char*& p, const char* end, int32_t& u, int32_t& it, double& v) {
    auto skip_ws = [&]() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; };
    auto parse_uint = [&](long& out) -> bool {
        skip_ws();
        const char* s = p;
        long x = 0;
        while (p < end && *p >= '0' && *p <= '9') { if (x < ((long)1 << 40)) x = x * 10 + (*p - '0'); ++p; }     // (saturates: an absurd token is an id out of range, not an overflow)
        if (p == s || (p < end && (*p == '.' || *p == '-' || *p == '+' || *p == 'e' || *p == 'E'))) { p = s; return false; }
        out = x;
        return true;
    };
    const char* save = p;
    long a, b;
    if (parse_uint(a) && parse_uint(b)) {
        skip_ws();
        const char* s = p;
        bool neg = false;
        if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
        long ip = 0; int nd = 0;
        while (p < end && *p >= '0' && *p <= '9' && nd < 15) { ip = ip * 10 + (*p - '0'); ++p; ++nd; }
        bool simple = nd > 0 || (p < end && *p == '.');
        double val = (double)ip;
        if (simple && p < end && *p == '.') {
            ++p;
            long fp = 0; int fd = 0;
            while (p < end && *p >= '0' && *p <= '9' && fd < 15) { fp = fp * 10 + (*p - '0'); ++p; ++fd; }
            static const double P10[16] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
            // exact only when the decimal is short; otherwise defer to strtod for correct rounding
            if (nd + fd > 15 || (p < end && *p >= '0' && *p <= '9')) simple = false;
            else if (fd > 0) val = (double)(ip * (long)P10[fd] + fp) / P10[fd];
        }
        // 1-based ids that do not fit int32 are malformed lines, not ids that wrap around into the valid range
        if (a < 1 || b < 1 || a > (long)INT32_MAX || b > (long)INT32_MAX) { p = save; return false; }
        if (simple && (p >= end || *p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == 0)) {
            u = (int32_t)(a - 1); it = (int32_t)(b - 1); v = neg ? -val : val;
            return true;
        }
        p = s;
        SlowCopy c(p, end);
        char* e;
        double dv = strtod(c.buf, &e);
        if (e == c.buf || (size_t)(e - c.buf) == sizeof(c.buf) - 1) { p = save; return false; }     // (a token that fills the copy is no rating)
        p += e - c.buf;
        u = (int32_t)(a - 1); it = (int32_t)(b - 1); v = dv;
        return true;
    }
    p = save;                                           // signs / odd formats in the ids: the libc path
    SlowCopy c(p, end);
    char* e;
    const char* q = c.buf;
    long i = strtol(q, &e, 10);
    if (e == q) return false;
    q = e;
    long j = strtol(q, &e, 10);
    if (e == q) return false;
    q = e;
    double dv = strtod(q, &e);
    if (e == q || (size_t)(e - c.buf) == sizeof(c.buf) - 1) return false;
    p += e - c.buf;
    if (i < INT32_MIN + 1 || j < INT32_MIN + 1 || i > (long)INT32_MAX || j > (long)INT32_MAX) return false;
    u = (int32_t)(i - 1); it = (int32_t)(j - 1); v = dv;
    return true;
}

// "%d %d %lf" per entry, 1-based ids (util.h:126,131; util.h:367-368); exactly nnz entries are read.
// Parsed by `threads` host threads: the file is cut at line boundaries, the non-blank lines of every piece are counted
// (one vectorisable pass; a piece that contains a line ending in white space is recounted exactly), then every piece parses
// into its slice of the (uninitialised) output arrays.
static int64_t nonblank_lines(const char* base, size_t a, size_t b) {
    // fast count: newlines, + 1 for an unterminated last line; exact unless some newline follows white space or another
    // newline (blank lines, "\r\n", trailing blanks) -- `odd` counts those, and any of them sends the piece to the exact loop
    int64_t nl = 0, odd = 0;
    if (b > a) { nl = base[a] == '\n'; odd = nl; }                        // (a newline at the start of a piece: a blank line)
    for (size_t q0 = a + 1; q0 < b; q0 += 1 << 15) {                      // (blocks with 32-bit counters and no loop-carried state: vectorises)
        const size_t q1 = std::min(b, q0 + ((size_t)1 << 15));
        const unsigned char* s = reinterpret_cast<const unsigned char*>(base);
        unsigned bn = 0, bo = 0;
        for (size_t q = q0; q < q1; ++q) {
            const unsigned isnl = s[q] == '\n', pws = (s[q - 1] == '\n') | (s[q - 1] == ' ') | (s[q - 1] == '\t') | (s[q - 1] == '\r');
            bn += isnl;
            bo += isnl & pws;
        }
        nl += bn; odd += bo;
    }
    if (odd == 0) {
        // an unterminated last line counts only if it has content ("1 2 3\n  " holds one entry, not two)
        bool tail = false;
        for (size_t q = b; q > a && base[q - 1] != '\n' && !tail; --q) tail = base[q - 1] != ' ' && base[q - 1] != '\t' && base[q - 1] != '\r';
        return nl + (tail ? 1 : 0);
    }
    int64_t n = 0; bool content = false;
    for (size_t q = a; q < b; ++q) {
        if (base[q] == '\n') { n += content; content = false; }
        else if (base[q] != ' ' && base[q] != '\t' && base[q] != '\r') content = true;
    }
    return n + (content ? 1 : 0);
}

// a rating file opened, cut into pieces at line boundaries, the entries before every piece counted
struct ScannedFile {
    TextFile tf;
    int T = 1;
    std::vector<size_t> cut;
    std::vector<int64_t> first;                            // first[t] = entries before piece t; first[T] = entries of the file
};
static int scan_ratings(const std::string& path, int threads, ScannedFile& sf) {
    int rc = sf.tf.open(path);
    if (rc != PCR_OK) return rc;
    const char* base = sf.tf.p;
    const size_t len = sf.tf.len;
    int T = std::max(1, std::min(threads, 64));
    if (len < (size_t)1 << 20) T = 1;
    sf.T = T;
    sf.cut.assign((size_t)T + 1, len);
    sf.cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        size_t c = std::max(sf.cut[t - 1], len * t / T);
        while (c < len && base[c] != '\n') ++c;
        sf.cut[t] = c < len ? c + 1 : len;
    }
    sf.first.assign((size_t)T + 1, 0);
    std::vector<int64_t> cnt(T, 0);
    run_pieces(T, T, [&](int, int64_t lo, int64_t hi) { for (int64_t t = lo; t < hi; ++t) cnt[t] = nonblank_lines(base, sf.cut[t], sf.cut[t + 1]); });
    for (int t = 0; t < T; ++t) sf.first[t + 1] = sf.first[t] + cnt[t];
    return PCR_OK;
}
// entries [0, n) of a scanned file into user / item / val (val may be null); returns the index of the first malformed entry, or -1
static int64_t parse_scanned(const ScannedFile& sf, int64_t n, int32_t* user, int32_t* item, double* val) {
    const int T = sf.T;
    const char* base = sf.tf.p;
    std::vector<int64_t> bad(T, -1);
    run_pieces(T, T, [&](int, int64_t lo, int64_t hi) {
        for (int64_t t = lo; t < hi; ++t) {
            const char* p = base + sf.cut[t];
            const char* end = base + sf.cut[t + 1];
            double dummy;
            for (int64_t z = sf.first[t]; z < sf.first[t + 1] && z < n; ++z)
                if (!parse_one(p, end, user[z], item[z], val ? val[z] : dummy)) { bad[t] = z; break; }
        }
    });
    for (int t = 0; t < T; ++t) if (bad[t] >= 0) return bad[t];
    return -1;
}

static int parse_ratings(const std::string& path, int64_t nnz, pcr_vec<int32_t>& user,
                         pcr_vec<int32_t>& item, pcr_vec<double>& val, int threads) {
    if (nnz < 0) { pcr_set_error(path + ": a negative rating count in meta"); return PCR_ERR_IO; }
    ScannedFile sf;
    int rc = scan_ratings(path, threads, sf);
    if (rc != PCR_OK) return rc;
    if (sf.first[sf.T] < nnz) {
        pcr_set_error(path + ": expected " + std::to_string(nnz) + " ratings, found " + std::to_string(sf.first[sf.T]));
        return PCR_ERR_IO;
    }
    user.resize((size_t)nnz); item.resize((size_t)nnz); val.resize((size_t)nnz);          // (only now: meta may promise any number)
    const int64_t bad = parse_scanned(sf, nnz, user.data(), item.data(), val.data());
    if (bad >= 0) { pcr_set_error(path + ": malformed rating line " + std::to_string(bad + 1)); return PCR_ERR_IO; }
    return PCR_OK;
}

// A rating file without a meta file beside it (pmf-predict.cpp:52-55 reads "user item rating" triples until fscanf fails):
// count = the file's non-blank lines; read = at most n entries by `threads` host threads, 0-based ids, ending at the first malformed
// entry (*n_read entries stand before it; the reference's `!= EOF` loop never ends there).
extern "C" int pcr_rating_file_count(const char* path, int64_t* n) {
    if (!path || !n) { pcr_set_error("pcr_rating_file_count: bad argument"); return PCR_ERR_ARG; }
    return guarded("pcr_rating_file_count", [&]() -> int {
        ScannedFile sf;
        int rc = scan_ratings(path, pcr_host_threads(), sf);
        if (rc != PCR_OK) return rc;
        *n = sf.first[sf.T];
        return PCR_OK;
    });
}
extern "C" int pcr_rating_file_read(const char* path, int threads, int64_t n, int32_t* user, int32_t* item, double* val, int64_t* n_read) {
    if (!path || n < 0 || !n_read || (n > 0 && (!user || !item))) { pcr_set_error("pcr_rating_file_read: bad argument"); return PCR_ERR_ARG; }
    if (threads <= 0) threads = pcr_host_threads();
    return guarded("pcr_rating_file_read", [&]() -> int {
        ScannedFile sf;
        int rc = scan_ratings(path, threads, sf);
        if (rc != PCR_OK) return rc;
        const int64_t m = std::min(n, sf.first[sf.T]);
        const int64_t bad = parse_scanned(sf, m, user, item, val);
        *n_read = bad >= 0 ? bad : m;
        return PCR_OK;
    });
}

extern "C" int pcr_dataset_load(const char* dir, pcr_dataset** out) { return pcr_dataset_load_mt(dir, 0, out); }

extern "C" int pcr_dataset_load_mt(const char* dir, int threads, pcr_dataset** out) {
    if (!dir || !out) { pcr_set_error("pcr_dataset_load: bad argument"); return PCR_ERR_ARG; }
    if (threads <= 0) threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::string d(dir);
    std::string metap = d + "/meta";
    FILE* fp = fopen(metap.c_str(), "r");
    if (!fp) { pcr_set_error("can't open " + metap + ": " + strerror(errno)); return PCR_ERR_IO; }
    long m = 0, n = 0, nnz = 0, tnnz = 0;
    char name[1024], tname[1024];
    bool have_test = false;
    if (fscanf(fp, "%ld %ld", &m, &n) != 2 || fscanf(fp, "%ld %1023s", &nnz, name) != 2) {
        fclose(fp);
        pcr_set_error(metap + ": expected 'm n' then 'nnz training_file'");
        return PCR_ERR_IO;
    }
    if (fscanf(fp, "%ld %1023s", &tnnz, tname) == 2) have_test = true;   // third line optional (util.cpp:18)
    fclose(fp);
    if (m < 0 || n < 0 || m > kMaxDim || n > kMaxDim) { pcr_set_error(metap + ": user / item counts must lie in [0, 2^31 - 2]"); return PCR_ERR_IO; }
    return guarded("pcr_dataset_load", [&]() -> int {
        // item / value arrays are parsed straight into the data set's CSR storage: for a file ordered by (user, item) they are
        // the CSR as they stand (build_train_csr adopts them)
        std::unique_ptr<pcr_dataset> ds(new pcr_dataset());
        pcr_vec<int32_t> u, tu;
        int rc = parse_ratings(d + "/" + name, nnz, u, ds->train.item, ds->train.val, threads);
        if (rc != PCR_OK) return rc;
        rc = build_train_csr(m, n, nnz, u.data(), ds->train.item.data(), ds->train.val.data(), ds->train, threads, true);
        if (rc != PCR_OK) return rc;
        pcr_vec<int32_t>().swap(u);
        if (have_test) {
            rc = parse_ratings(d + "/" + tname, tnnz, tu, ds->test.item, ds->test.val, threads);
            if (rc != PCR_OK) return rc;
            rc = check_test_ids(n, tnnz, tu.data(), ds->test.item.data(), threads);
            if (rc != PCR_OK) return rc;
        } else {
            tnnz = 0;
        }
        build_test_csr(m, n, tnnz, tu.data(), ds->test.item.data(), ds->test.val.data(), ds->test, threads, true);
        ds->tnnz_file = tnnz;
        *out = ds.release();
        return PCR_OK;
    });
}

// ---- binary side-car cache of the converted data set
namespace {
struct CacheHeader {
    char magic[8];                 // "PCRCACH1"
    int64_t d1, d2, nnz, tnnz, tnnz_file;
    int64_t stamp[6];              // size, mtime (ns) of meta, training file, test file at the time of the parse (0: not recorded)
};
const char kCacheMagic[8] = {'P', 'C', 'R', 'C', 'A', 'C', 'H', '1'};

bool file_stamp(const std::string& p, int64_t* size, int64_t* mtime) {
    struct stat sb;
    if (stat(p.c_str(), &sb) != 0) return false;
    *size = (int64_t)sb.st_size;
    *mtime = (int64_t)sb.st_mtim.tv_sec * 1000000000LL + (int64_t)sb.st_mtim.tv_nsec;
    return true;
}
// stamps of <dir>/meta and the rating files it names; false if meta is unreadable
bool dir_stamps(const std::string& d, int64_t stamp[6]) {
    for (int i = 0; i < 6; ++i) stamp[i] = 0;
    FILE* fp = fopen((d + "/meta").c_str(), "r");
    if (!fp) return false;
    long m = 0, n = 0, nnz = 0, tnnz = 0;
    char name[1024], tname[1024];
    const bool ok = fscanf(fp, "%ld %ld", &m, &n) == 2 && fscanf(fp, "%ld %1023s", &nnz, name) == 2;
    const bool have_test = ok && fscanf(fp, "%ld %1023s", &tnnz, tname) == 2;
    fclose(fp);
    if (!ok || !file_stamp(d + "/meta", &stamp[0], &stamp[1]) || !file_stamp(d + "/" + name, &stamp[2], &stamp[3])) return false;
    if (have_test && !file_stamp(d + "/" + tname, &stamp[4], &stamp[5])) return false;
    return true;
}

// n bytes between a file and memory by the host threads: pread / pwrite of disjoint pieces (both are thread-safe on one descriptor)
bool par_io(int fd, off_t off, void* mem, size_t n, bool write, int threads) {
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(threads, 64), n / ((size_t)8 << 20) + 1));
    std::vector<char> ok((size_t)T, 1);
    run_pieces((int64_t)n, T, [&](int t, int64_t lo, int64_t hi) {
        char* p = static_cast<char*>(mem);
        while (lo < hi) {
            const ssize_t g = write ? pwrite(fd, p + lo, (size_t)(hi - lo), off + lo) : pread(fd, p + lo, (size_t)(hi - lo), off + lo);
            if (g < 0 && errno == EINTR) continue;
            if (g <= 0) { ok[(size_t)t] = 0; return; }
            lo += g;
        }
    });
    for (char c : ok) if (!c) return false;
    return true;
}

// Layout: header | train.index | train.item | train.val | test.index | test.item | test.val.  Written and read by the host
// threads in disjoint pieces (a 1.2 GB cache through one fread / fwrite was SLOWER than parsing the text it caches).
int save_cache(const pcr_dataset* ds, const char* path, const int64_t stamp[6]) {
    if (!ds || !path) { pcr_set_error("pcr_dataset_save_cache: bad argument"); return PCR_ERR_ARG; }
    const std::string tmp = std::string(path) + ".tmp";
    const int fd = ::open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0644);
    if (fd < 0) { pcr_set_error("can't open " + tmp + ": " + strerror(errno)); return PCR_ERR_IO; }
    CacheHeader h;
    memcpy(h.magic, kCacheMagic, 8);
    h.d1 = ds->train.d1; h.d2 = ds->train.d2; h.nnz = ds->train.nnz(); h.tnnz = ds->test.nnz(); h.tnnz_file = ds->tnnz_file;
    for (int i = 0; i < 6; ++i) h.stamp[i] = stamp ? stamp[i] : 0;
    const int threads = pcr_host_threads();
    off_t off = 0;
    bool ok = par_io(fd, off, &h, sizeof(h), true, 1);
    off += sizeof(h);
    auto put = [&](const void* p, size_t bytes) { if (ok && bytes) ok = par_io(fd, off, const_cast<void*>(p), bytes, true, threads); off += (off_t)bytes; };
    put(ds->train.index.data(), ds->train.index.size() * 8); put(ds->train.item.data(), ds->train.item.size() * 4); put(ds->train.val.data(), ds->train.val.size() * 8);
    put(ds->test.index.data(), ds->test.index.size() * 8); put(ds->test.item.data(), ds->test.item.size() * 4); put(ds->test.val.data(), ds->test.val.size() * 8);
    ok = (::close(fd) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); pcr_set_error(std::string("can't write ") + path); return PCR_ERR_IO; }
    return PCR_OK;
}
int load_cache(const char* path, const int64_t want[6], pcr_dataset** out) {
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { pcr_set_error(std::string("can't open ") + path + ": " + strerror(errno)); return PCR_ERR_IO; }
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    CacheHeader h;
    auto fail = [&](const char* why) { pcr_set_error(std::string(path) + ": " + why); return PCR_ERR_IO; };
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(h) || !par_io(fd, 0, &h, sizeof(h), false, 1) || memcmp(h.magic, kCacheMagic, 8) != 0)
        return fail("not a data set cache of this version");
    if (h.d1 < 0 || h.d2 < 0 || h.nnz < 0 || h.tnnz < 0 || h.d1 > kMaxDim || h.d2 > kMaxDim) return fail("corrupt header");
    {   // the header must describe exactly this file before anything is allocated from it
        const __int128 want_bytes = (__int128)sizeof(h) + 2 * (__int128)(h.d1 + 1) * 8 + (__int128)(h.nnz + h.tnnz) * 12;
        if ((__int128)sb.st_size != want_bytes) return fail("truncated or corrupt (size does not match the header)");
    }
    if (want) for (int i = 0; i < 6; ++i) if (h.stamp[i] != want[i]) return fail("stale (the text files changed)");
    const int threads = pcr_host_threads();
    std::unique_ptr<pcr_dataset> ds(new pcr_dataset);
    ds->train.d1 = ds->test.d1 = h.d1; ds->train.d2 = ds->test.d2 = h.d2; ds->tnnz_file = h.tnnz_file;
    ds->train.index.resize((size_t)h.d1 + 1); ds->train.item.resize((size_t)h.nnz); ds->train.val.resize((size_t)h.nnz);
    ds->test.index.resize((size_t)h.d1 + 1); ds->test.item.resize((size_t)h.tnnz); ds->test.val.resize((size_t)h.tnnz);
    off_t off = sizeof(h);
    bool ok = true;
    auto get = [&](void* p, size_t bytes) { if (ok && bytes) ok = par_io(fd, off, p, bytes, false, threads); off += (off_t)bytes; };
    get(ds->train.index.data(), ds->train.index.size() * 8); get(ds->train.item.data(), ds->train.item.size() * 4); get(ds->train.val.data(), ds->train.val.size() * 8);
    get(ds->test.index.data(), ds->test.index.size() * 8); get(ds->test.item.data(), ds->test.item.size() * 4); get(ds->test.val.data(), ds->test.val.size() * 8);
    if (!ok) return fail("truncated");
    const std::vector<int64_t>&ti = ds->train.index, &xi = ds->test.index;
    if (ti.front() != 0 || ti.back() != h.nnz || xi.front() != 0 || xi.back() != h.tnnz) return fail("corrupt row pointers");
    // consistency of everything an index will be taken from, user ranges side by side: 0 = fine, else the first kind of damage
    const int T = pieces_for(h.nnz + h.d1, threads);
    std::vector<int> bad((size_t)T, 0);
    run_pieces(h.d1, T, [&](int t, int64_t lo, int64_t hi) {
        for (int64_t u = lo; u < hi; ++u) {
            if (ti[u + 1] < ti[u] || xi[u + 1] < xi[u] || ti[u + 1] > h.nnz || xi[u + 1] > h.tnnz || ti[u] < 0 || xi[u] < 0) { bad[(size_t)t] = 1; return; }
            for (int64_t z = ti[u]; z < ti[u + 1]; ++z) {
                const int32_t j = ds->train.item[(size_t)z];
                if (j < 0 || j >= h.d2) { bad[(size_t)t] = 2; return; }
                if (z > ti[u] && j <= ds->train.item[(size_t)z - 1]) { bad[(size_t)t] = 3; return; }
            }
            for (int64_t z = xi[u]; z < xi[u + 1]; ++z)
                if (ds->test.item[(size_t)z] < 0 || ds->test.item[(size_t)z] >= h.d2) { bad[(size_t)t] = 2; return; }
        }
    });
    for (int b : bad)
        if (b) return fail(b == 1 ? "corrupt row pointers" : b == 2 ? "item id out of range" : "items of a user not ascending");
    *out = ds.release();
    return PCR_OK;
}
}  // namespace

extern "C" int pcr_dataset_save_cache(const pcr_dataset* ds, const char* path) { return save_cache(ds, path, nullptr); }
extern "C" int pcr_dataset_load_cache(const char* path, pcr_dataset** out) {
    if (!path || !out) { pcr_set_error("pcr_dataset_load_cache: bad argument"); return PCR_ERR_ARG; }
    return guarded("pcr_dataset_load_cache", [&]() -> int { return load_cache(path, nullptr, out); });
}
extern "C" int pcr_dataset_load_cached(const char* dir, int threads, const char* cache, pcr_dataset** out) {
    if (!dir || !cache || !out) { pcr_set_error("pcr_dataset_load_cached: bad argument"); return PCR_ERR_ARG; }
    int64_t stamp[6];
    const bool have = dir_stamps(dir, stamp);
    if (have && guarded("pcr_dataset_load_cached", [&]() -> int { return load_cache(cache, stamp, out); }) == PCR_OK) return PCR_OK;
    int rc = pcr_dataset_load_mt(dir, threads, out);
    if (rc != PCR_OK) return rc;
    if (have) (void)save_cache(*out, cache, stamp);      // best effort: a read-only data directory is not an error
    return PCR_OK;
}

extern "C" void pcr_dataset_free(pcr_dataset* ds) { delete ds; }

extern "C" int pcr_dataset_dims(const pcr_dataset* ds, int64_t* d1, int64_t* d2, int64_t* nnz, int64_t* tnnz) {
    if (!ds) { pcr_set_error("null dataset"); return PCR_ERR_ARG; }
    if (d1) *d1 = ds->train.d1;
    if (d2) *d2 = ds->train.d2;
    if (nnz) *nnz = ds->train.nnz();
    if (tnnz) *tnnz = ds->test.nnz();
    return PCR_OK;
}

extern "C" int pcr_dataset_csr(const pcr_dataset* ds, int which, int64_t* index, int64_t* item, double* val) {
    if (!ds || (which != 0 && which != 1)) { pcr_set_error("pcr_dataset_csr: bad argument"); return PCR_ERR_ARG; }
    const PcrCsr& X = which == 0 ? ds->train : ds->test;
    if (index) std::copy(X.index.begin(), X.index.end(), index);
    if (item) for (size_t z = 0; z < X.item.size(); ++z) item[z] = X.item[z];
    if (val) std::copy(X.val.begin(), X.val.end(), val);
    return PCR_OK;
}

// ------------------------------------------------------------------------------------------
// levels
// ------------------------------------------------------------------------------------------

// run fn(t, lo, hi) for nthreads contiguous pieces of [0, n) (host set-up work: levels, tile-major CSC)
void pcr_parallel_ranges(int64_t n, int nthreads, const std::function<void(int, int64_t, int64_t)>& fn) {
    nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / 4096 + 1));
    if (nthreads == 1) { fn(0, 0, n); return; }
    run_workers(nthreads, [&](int t) { fn(t, n * t / nthreads, n * (t + 1) / nthreads); });
}
// fn(task) for task = 0 .. ntasks - 1, handed out one by one to nthreads workers (uneven tasks); exceptions are carried back
void pcr_parallel_tasks(int ntasks, int nthreads, const std::function<void(int)>& fn) {
    std::atomic<int> next{0};
    run_workers(std::max(1, std::min(nthreads, ntasks)), [&](int) { for (;;) { const int t = next.fetch_add(1); if (t >= ntasks) break; fn(t); } });
}
int pcr_host_threads() {
    // (the cgroup quota of a container is not visible through hardware_concurrency: cap at 16, the loader's default too)
    return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
}

// Three passes, the outer two over user ranges in parallel: (1) per user the sorted distinct keys and each rating's level;
// (2) prefix sums of the per-user level counts; (3) the cumulative per-level counts and the level keys.
// Pass 1 per user: rating sets have a handful of integer levels, so when the rounded keys of a user span fewer than 64 integers
// (and, for PrimalCR's raw keys, every rating IS an integer) the level of a rating is a popcount in a 64-bit presence mask -- two
// streaming passes over the ratings, no searching; otherwise insertion into a small sorted vector (<= 64 distinct values), else
// sort + unique.
// lround() without the libm call: truncate, then step away from zero when the (exactly representable) remainder is at least a half
static inline long fast_lround(double v) {
    if (!(v > -4.0e15 && v < 4.0e15)) return lround(v);
    const long t = (long)v;
    const double f = v - (double)t;
    return f >= 0.5 ? t + 1 : f <= -0.5 ? t - 1 : t;
}

int pcr_build_levels(const PcrCsr& X, int64_t u0, int64_t u1, int solver_type, PcrLevels& out, std::string& err) {
    const int64_t z0 = X.index[u0], z1 = X.index[u1], nu = u1 - u0;
    out.level.resize((size_t)(z1 - z0));
    out.run_ofs.assign((size_t)(nu + 1), 0);
    out.run_start.clear();
    out.lev_val.clear();
    out.max_levels = 0;
    const int nth = pcr_host_threads();
    std::vector<int64_t> bad(nth, -1);
    std::vector<int> maxlev(nth, 0);
    std::vector<char> allint(nth, 1);
    const bool pp = solver_type == PCR_SOLVER_PCRPP;
    // per user: the presence mask + its base (fast path) or nothing (mask 0: general path, redone in pass 3 for the keys)
    std::vector<uint64_t> umask((size_t)nu, 0);
    std::vector<int64_t> ubase((size_t)nu, 0);
    const double* val = X.val.data();
    pcr_parallel_ranges(nu, nth, [&](int t, int64_t lo, int64_t hi) {
        std::vector<double> uniq, keys;
        std::vector<long> ikey;
        for (int64_t ui = lo; ui < hi; ++ui) {
            const int64_t a = X.index[u0 + ui], b = X.index[u0 + ui + 1];
            if (b == a) { out.run_ofs[ui + 1] = 1; continue; }
            long kmin = LONG_MAX, kmax = LONG_MIN;
            bool ints = true;
            ikey.resize((size_t)(b - a));                          // the rounded keys once (the user's ratings stay in cache)
            for (int64_t z = a; z < b; ++z) {
                const double v = val[z];
                const long k = fast_lround(v);
                ikey[(size_t)(z - a)] = k;
                kmin = std::min(kmin, k); kmax = std::max(kmax, k);
                ints = ints && (double)k == v;
            }
            if (!ints) allint[t] = 0;
            if ((pp || ints) && kmax - kmin < 64 && kmin > LONG_MIN / 2 && kmax < LONG_MAX / 2) {
                uint64_t m = 0;
                for (int64_t z = a; z < b; ++z) m |= (uint64_t)1 << (ikey[(size_t)(z - a)] - kmin);
                for (int64_t z = a; z < b; ++z) {
                    const int sh = (int)(ikey[(size_t)(z - a)] - kmin);
                    out.level[z - z0] = (uint16_t)__builtin_popcountll(m & (((uint64_t)1 << sh) - 1));
                }
                const int T = __builtin_popcountll(m);
                umask[ui] = m; ubase[ui] = kmin;
                maxlev[t] = std::max(maxlev[t], T);
                out.run_ofs[ui + 1] = T + 1;
                continue;
            }
            uniq.clear();
            bool small = true;
            for (int64_t z = a; z < b && small; ++z) {
                const double k = pp ? (double)fast_lround(val[z]) : val[z];
                auto it = std::lower_bound(uniq.begin(), uniq.end(), k);
                if (it == uniq.end() || *it != k) { if (uniq.size() >= 64) small = false; else uniq.insert(it, k); }
            }
            if (!small) {
                keys.resize((size_t)(b - a));
                for (int64_t z = a; z < b; ++z) keys[z - a] = pp ? (double)fast_lround(val[z]) : val[z];
                uniq = keys;
                std::sort(uniq.begin(), uniq.end());
                uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            }
            const int T = (int)uniq.size();
            if (T > 65535) { if (bad[t] < 0) bad[t] = u0 + ui; continue; }
            maxlev[t] = std::max(maxlev[t], T);
            out.run_ofs[ui + 1] = T + 1;
            for (int64_t z = a; z < b; ++z) {
                const double k = pp ? (double)fast_lround(val[z]) : val[z];
                out.level[z - z0] = (uint16_t)(std::lower_bound(uniq.begin(), uniq.end(), k) - uniq.begin());
            }
        }
    });
    out.integer_valued = true;
    for (int t = 0; t < nth; ++t) {
        if (bad[t] >= 0) { err = "user " + std::to_string(bad[t]) + " has more than 65535 distinct rating levels"; return PCR_ERR_UNSUPPORTED; }
        out.max_levels = std::max(out.max_levels, maxlev[t]);
        out.integer_valued = out.integer_valued && allint[t];
    }
    for (int64_t ui = 0; ui < nu; ++ui) out.run_ofs[ui + 1] += out.run_ofs[ui];
    out.run_start.assign((size_t)out.run_ofs[nu], 0);
    out.lev_val.assign((size_t)out.run_ofs[nu], 0.0);
    pcr_parallel_ranges(nu, nth, [&](int, int64_t lo, int64_t hi) {
        for (int64_t ui = lo; ui < hi; ++ui) {
            const int64_t a = X.index[u0 + ui], b = X.index[u0 + ui + 1];
            int32_t* rs = out.run_start.data() + out.run_ofs[ui];
            double* lvv = out.lev_val.data() + out.run_ofs[ui];
            const int T = (int)(out.run_ofs[ui + 1] - out.run_ofs[ui]) - 1;
            for (int64_t z = a; z < b; ++z) rs[out.level[z - z0] + 1]++;
            for (int l = 0; l < T; ++l) rs[l + 1] += rs[l];
            if (umask[ui]) {
                int l = 0;
                for (uint64_t m = umask[ui]; m; m &= m - 1) lvv[l++] = (double)(ubase[ui] + __builtin_ctzll(m));
            } else {
                for (int64_t z = a; z < b; ++z) lvv[out.level[z - z0]] = pp ? (double)fast_lround(val[z]) : val[z];
            }
        }
    });
    return PCR_OK;
}

extern "C" int64_t pcr_dataset_count_pairs(const pcr_dataset* ds, int solver_type) {
    if (!ds) return -1;
    const PcrCsr& X = ds->train;
    PcrLevels lv;
    std::string err;
    if (pcr_build_levels(X, 0, X.d1, solver_type, lv, err) != PCR_OK) { pcr_set_error(err); return -1; }
    int64_t total = 0;
    for (int64_t u = 0; u < X.d1; ++u) {
        int64_t n = X.index[u + 1] - X.index[u], same = 0;
        for (int64_t q = lv.run_ofs[u]; q + 1 < lv.run_ofs[u + 1]; ++q) {
            int64_t c = lv.run_start[q + 1] - lv.run_start[q];
            same += c * c;
        }
        total += (n * n - same) / 2;
    }
    return total;
}

// ------------------------------------------------------------------------------------------
// model file (pmf-train.cpp:297-310, util.cpp:30-79)
// ------------------------------------------------------------------------------------------

extern "C" int pcr_model_save(const char* path, const double* U, int64_t d1, const double* V, int64_t d2, int64_t k) {
    FILE* fp = fopen(path, "wb");
    if (!fp) { pcr_set_error(std::string("can't open output file ") + path); return PCR_ERR_IO; }
    long hdr[2];
    bool ok = true;
    hdr[0] = (long)d1; hdr[1] = (long)k;
    ok &= fwrite(hdr, sizeof(long), 2, fp) == 2;
    ok &= fwrite(U, sizeof(double), (size_t)(d1 * k), fp) == (size_t)(d1 * k);
    hdr[0] = (long)d2; hdr[1] = (long)k;
    ok &= fwrite(hdr, sizeof(long), 2, fp) == 2;
    ok &= fwrite(V, sizeof(double), (size_t)(d2 * k), fp) == (size_t)(d2 * k);
    ok &= fclose(fp) == 0;
    if (!ok) { pcr_set_error(std::string("short write to ") + path); return PCR_ERR_IO; }
    return PCR_OK;
}

extern "C" int pcr_model_load(const char* path, int64_t* d1, int64_t* d2, int64_t* k, double* U, double* V) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { pcr_set_error(std::string("can't open model file ") + path); return PCR_ERR_IO; }
    long hdr[2];
    int rc = PCR_OK;
    struct stat sb;
    const bool have_size = fstat(fileno(fp), &sb) == 0;
    do {
        if (fread(hdr, sizeof(long), 2, fp) != 2 || hdr[0] < 0 || hdr[1] < 0) { rc = PCR_ERR_IO; break; }
        long m1 = hdr[0], kk = hdr[1];
        // (a corrupt header must not turn into a 2^60-element read or, in a caller that sizes its buffers from the query call, an
        // allocation of that size: the first matrix has to fit the file)
        if (!have_size || (__int128)m1 * kk * 8 + 32 > (__int128)sb.st_size) { rc = PCR_ERR_IO; break; }
        if (U) { if (fread(U, sizeof(double), (size_t)(m1 * kk), fp) != (size_t)(m1 * kk)) { rc = PCR_ERR_IO; break; } }
        else fseek(fp, (long)sizeof(double) * m1 * kk, SEEK_CUR);
        if (fread(hdr, sizeof(long), 2, fp) != 2 || hdr[1] != kk || hdr[0] < 0) { rc = PCR_ERR_IO; break; }
        long m2 = hdr[0];
        if ((__int128)(m1 + m2) * kk * 8 + 32 != (__int128)sb.st_size) { rc = PCR_ERR_IO; break; }
        if (V && fread(V, sizeof(double), (size_t)(m2 * kk), fp) != (size_t)(m2 * kk)) { rc = PCR_ERR_IO; break; }
        if (d1) *d1 = m1;
        if (d2) *d2 = m2;
        if (k) *k = kk;
    } while (0);
    fclose(fp);
    if (rc != PCR_OK) pcr_set_error(std::string("malformed model file ") + path);
    return rc;
}

// ------------------------------------------------------------------------------------------
// multi-GPU partition: contiguous user ranges balanced by nnz
// ------------------------------------------------------------------------------------------

extern "C" int pcr_partition_users(const int64_t* index, int64_t d1, int nparts, int64_t* bounds) {
    if (!index || !bounds || nparts < 1 || d1 < 0) { pcr_set_error("pcr_partition_users: bad argument"); return PCR_ERR_ARG; }
    int64_t nnz = index[d1];
    bounds[0] = 0;
    for (int p = 1; p < nparts; ++p) {
        // first user boundary whose prefix nnz reaches p/nparts of the total
        double target = (double)nnz * p / nparts;
        int64_t lo = bounds[p - 1], hi = d1;
        while (lo < hi) {
            int64_t mid = (lo + hi) / 2;
            if ((double)index[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (nnz == 0) lo = d1 * p / nparts;
        bounds[p] = std::max(lo, bounds[p - 1]);
    }
    bounds[nparts] = d1;
    return PCR_OK;
}
