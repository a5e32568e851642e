// pcr_solver.hip -- device solver behind the C ABI of include/primalcr.h.
//
// Host orchestration of the PrimalCR++ / PrimalCR training loop (pcrpp.cpp:841-901,
// pcr.cpp:616-704) over the HIP kernels of pcr_kernels.h.  One process per GPU; users are
// sharded across ranks (contiguous, nnz-balanced), V and the CG vectors are replicated, the
// V-gradient and every Hessian-vector product are combined with an RCCL all-reduce.
//
// There is NO CPU compute path in this file: every [device] entry point fails with
// PCR_ERR_DEVICE when HIP is unusable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pcr_host.h"
#include "pcr_kernels.h"
#include "pcr_gram.h"
#include "pcr_newton.h"
#include "pcr_vblock.h"
#include "pcr_p2p.h"

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pcr_set_error(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
            return PCR_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)
#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t e_ = (expr);                                                              \
        if (e_ != ncclSuccess) {                                                               \
            pcr_set_error(std::string(#expr) + ": " + ncclGetErrorString(e_));                 \
            return PCR_ERR_COMM;                                                               \
        }                                                                                      \
    } while (0)
#define RC(expr) do { int rc_ = (expr); if (rc_ != PCR_OK) return rc_; } while (0)

static inline int host_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
#include "pcr_plan.h"
#include "pcr_plan_dev.h"

// device buffer with RAII
template <typename X>
struct DBuf {
    X* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        free();
        n = count;
        if (count == 0) count = 1;
        HIPCHK(hipMalloc((void**)&p, count * sizeof(X)));
        return PCR_OK;
    }
    int upload(const std::vector<X>& h, hipStream_t st) {
        RC(alloc(h.size()));
        (void)st;
        if (!h.empty()) HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(X), hipMemcpyHostToDevice));
        return PCR_OK;
    }
    int upload_n(const X* h, size_t count) {
        RC(alloc(count));
        if (count) HIPCHK(hipMemcpy(p, h, count * sizeof(X), hipMemcpyHostToDevice));
        return PCR_OK;
    }
    void free() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { free(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DBuf() { free(); }
};

// users of one CSR grouped by length class; each class has its own workgroup size
struct Bin {
    int block = 64;
    bool big = false;
    int K = 1;           // workgroups per user (k_ustep clusters)
    int ugrid = 0;       // k_ustep grid of this bin
    int scratch_ofs = 0; // first global-scratch slice of this bin (big bins that run concurrently must not share slices)
    int cap = 0;         // longest user in the bin
    int limit = 0;       // upper length bound of the class (0: none)
    int rcap = 0;        // k_ustep: rows of V a workgroup keeps resident in LDS
    int unr = 4;         // k_ustep: rows in flight per lane group (8: latency-bound class, one workgroup per CU)
    bool gram = false;   // k_ustep_gram: the dual (Gram-matrix, MFMA) form for users with few ratings
    int wcap = 0;        // k_ustep: 16-bit window entries cached in LDS (cap * ws, or 0)
    int sym = 0;         // k_ustep: symbol id (template parameter CLS) among the classes that run the same workgroup form
    int max_lev = 0;
    int64_t nnz = 0;     // ratings of the users in the bin
    std::vector<int32_t> users;
    DBuf<int32_t> d_users;
};
// length classes: one wave for short users, 256 threads up to 512 ratings, 512 threads up to 4096
// (all with the user's block in LDS), longer users through global scratch.  512 rather than 1024
// threads for the top classes: k_ustep needs more than the 128 VGPRs a 512-thread block may use.
static const int BIN_LIMIT[3] = {128, 512, 4096};
static const int BIN_BLOCK[4] = {64, 256, 512, 512};
static const int GRAM_DEFAULT_CAP = 0;       // default length bound of the dual-form U-step class (0: off; pcr_tune "ustep_gram")

struct ProfSlot {
    int64_t ratings = -1, users = -1;      // what one launch covers (-1: the whole shard)
    int64_t seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0;
    int64_t n = 0;
};

struct pcr_solver {
    virtual ~pcr_solver() {}
    virtual int set_factors(const double* U, const double* V, bool local) = 0;
    virtual int get_factors(double* U, double* V, bool local) = 0;
    virtual int comp_m(double* m_out) = 0;
    virtual int objective(double* obj) = 0;
    virtual int obtain_g(double* g) = 0;
    virtual int compute_Ha(const double* a, double* Ha) = 0;
    virtual int solve_delta(const double* g, double* delta, int* iters) = 0;
    virtual int update_V(double* now_obj, int* info) = 0;
    virtual int update_U(double* now_obj, int64_t* info) = 0;
    virtual int evaluate(int which, int ndcg_k, double* err, double* ndcg) = 0;
    virtual int train(pcr_log_fn log, void* ctx, pcr_iter_stats* hist) = 0;
    virtual int iterate_abi(int n, pcr_iter_stats* out) = 0;
    virtual int comm_init(const void* id) = 0;
    virtual int comm_init_p2p(const char* shm_name) = 0;
    virtual void comm_abort() = 0;
    virtual int comm_nranks() = 0;
    virtual int sync() = 0;
    virtual std::string ustep_classes() = 0;                       // comma-separated profile slot names of the U-step length classes
    virtual int class_rows(const std::string& slot, double* v) = 0; // rows of V that class has gathered so far (pcr_tune "count_rows")
    int64_t first_user = 0, n_users = 0, nnz_local = 0;
    double ustep_rows = 0.0;      // rows of V gathered by all U steps so far (all ranks); pcr_solver_counter("ustep_row_gathers")
    bool prof_on = false;
    bool local_only = false;      // nranks > 1 without a communicator: entry points return this shard's partials
    int prof_period = 1;          // time every prof_period-th launch of each slot
    std::map<std::string, ProfSlot> prof;
    std::vector<std::pair<std::string, double>> setup_ms;   // wall time of the phases of pcr_solver_create, in order (pcr_solver_counter "setup_ms/<i>", pcr_solver_setup_phase)
    virtual int prof_resolve() = 0;
    virtual void prof_prewarm(int n) = 0;
};

// launch knobs: pcr_tune() values read once when the solver is created (include/primalcr.h lists them)
struct Tune {
    int lanes = 0, spmm_chunk = 0, spmm_tiles = 0, sddmm_csc = -1, ustep_mode = 0, cluster_k = 4, cluster_users = 0, window_cache = 1,
        prepare_merged = -1, pipeline = 1, debug = 0, fault_cluster_member = 0, ustep_gram = -1, count_rows = 0, allreduce_chunks = 0,
        resort_window = 8, p2p_ll = 16, p2p_timeout_ms = 20000, p2p_queue_budget = 0, fault_p2p_skip = 0, fault_p2p_coarse = 0, win16 = 1, ustep_win_lds = 1, plan_key64 = 0, ustep_newton = 0, vblock_users = 0;
    std::string ubins;
    void read() {
        lanes = pcr_tune_int("lanes", 0); spmm_chunk = pcr_tune_int("spmm_chunk", 0); spmm_tiles = pcr_tune_int("spmm_tiles", 0);
        sddmm_csc = pcr_tune_int("sddmm_csc", -1); ustep_mode = pcr_tune_int("ustep_mode", 0); cluster_k = pcr_tune_int("cluster_k", 4);
        cluster_users = pcr_tune_int("cluster_users", 0); window_cache = pcr_tune_int("window_cache", 1);
        prepare_merged = pcr_tune_int("prepare_merged", -1);
        pipeline = pcr_tune_int("pipeline", 1); debug = pcr_tune_int("debug", 0); fault_cluster_member = pcr_tune_int("fault_cluster_member", 0);
        ustep_gram = pcr_tune_int("ustep_gram", -1); count_rows = pcr_tune_int("count_rows", 0);
        allreduce_chunks = pcr_tune_int("allreduce_chunks", 0);
        resort_window = pcr_tune_int("resort_window", 8); p2p_ll = pcr_tune_int("p2p_ll", 16);
        p2p_timeout_ms = pcr_tune_int("p2p_timeout_ms", 20000); fault_p2p_skip = pcr_tune_int("fault_p2p_skip", 0);
        fault_p2p_coarse = pcr_tune_int("fault_p2p_coarse", 0); p2p_queue_budget = pcr_tune_int("p2p_queue_budget", 0);
        win16 = pcr_tune_int("win16", 1); ustep_win_lds = pcr_tune_int("ustep_win_lds", 1); plan_key64 = pcr_tune_int("plan_key64", 0); ustep_newton = pcr_tune_int("ustep_newton", 0); vblock_users = pcr_tune_int("vblock_users", 0);
        ubins.clear(); pcr_tune_get("ubins", &ubins);
    }
};

template <typename T>
struct Solver final : pcr_solver {
    pcr_params prm;
    Tune tune;
    int rank = 0, nranks = 1;
    int64_t d1 = 0, d2 = 0, tnnz_file = 0;
    Geo geo;
    hipStream_t st = nullptr;
    static constexpr int NSIDE = 12;
    hipStream_t side[NSIDE] = {};                                 // length bins run concurrently
    hipEvent_t ev_fork = nullptr;
    hipStream_t hi = nullptr;                                     // high priority: the cluster class of the U step
    hipEvent_t ev_hi = nullptr;
    // Streams that really run side by side.  HIP multiplexes its streams onto a few hardware queues (4 by default) in
    // creation order across the whole process, and two streams on one queue serialise: which of ours collide depends on
    // what else the process created.  pick_lanes() measures it once and keeps up to 4 mutually independent streams
    // (lane[0] = the solver's stream); concurrent length classes are placed on lanes only.
    static constexpr int MAXLANE = 8;
    hipStream_t lane[MAXLANE] = {};
    hipEvent_t ev_lane[MAXLANE] = {};
    int nlane = 1;
    int pipe_lanes = 0;                                           // side lanes measured to share the solver's stream's command-processor pipe (they are last in lane[])
    bool hi_on_solver_pipe = false;
    std::vector<hipStream_t> hi_spare;                            // high-priority streams that landed on the solver's pipe and were replaced
    ncclComm_t comm = nullptr;
    std::unique_ptr<P2PComm> p2p;                                 // the direct peer-to-peer alternative (pcr_p2p.h)
    bool single() const { return (nranks == 1 && !comm && !p2p) || local_only; }     // no exchange step: one shard, or shard-local mode
    int ncu = 256;

    // ---- training shard
    Shard<T> sh;
    DBuf<int64_t> d_uptr, d_runofs;
    DBuf<int32_t> d_sidx;
    DBuf<double> d_objr;
    DBuf<int32_t> d_item, d_c2r, d_runstart, d_sitem, d_cuser, d_crow, d_ruser, d_slot_base, d_item_slot, d_chunk_ptr, d_slot_id;
    DBuf<int2> d_blk_chunks;                      // k_spmm: first chunk and chunk count of every workgroup
    // pcr_tune("vblock_users"): the blocked-user V step (pcr_vblock.h): the block's users (padded to 32 with -1), the dense
    // (user, item) -> CSR position index, the per-user exclusion flags of the sparse plan
    DBuf<int32_t> d_blk_user, d_cpos_dense;
    DBuf<unsigned char> d_excl;
    int vblock_nbp = 0;
    // pcr_tune("ustep_newton"): the users k_unewton covers, its directions (nu x ld, NaN = none), its prefix tables
    DBuf<int32_t> d_newton_users;
    DBuf<double> d_dir;
    DBuf<char> d_newton_scratch;
    int newton_n = 0, newton_cap = 0, newton_rs = 0, newton_grid = 0;
    size_t newton_stride = 0;
    DBuf<int32_t> d_cuf;                          // k_spmm: user id | new-item flag per CSC entry
    int spmm_blocks = 0, spmm_tiles = 1;
    // Item ranges of the SpMM (N > 1, wide item tables): k_spmm / k_spmm_fin run range by range and the all-reduce of a finished
    // range overlaps the SpMM of the next one (stream ar_st).  Range r = items [rng_item[r], rng_item[r+1]): the same cut on
    // every rank (it depends on d2 alone); its workgroups are blk_chunks[rng_blk[r] .. rng_blk[r+1]).
    int n_rng = 1;
    std::vector<int64_t> rng_item;
    std::vector<int> rng_blk;
    hipStream_t ar_st = nullptr;
    std::vector<hipEvent_t> ev_rng;
    hipEvent_t ev_ar = nullptr;
    bool sddmm_csc = false;                       // the CG's SDDMM walks the SpMM's tile-major CSC (item table beyond the L2s)
    DBuf<T> d_slab;                               // k_spmm partial rows, one per (chunk, item) incidence
    bool sweep_pf4 = false;                       // sweeps keep four rounds of per-rating loads in flight (large shards)
    int spmm_chunk = 128;
    int sddmm_tile = 0;                           // ratings per lane group of k_sddmm (0 = not chosen yet)
    DBuf<uint16_t> d_lvl, d_slvl;
    DBuf<unsigned char> d_rhint;                  // k_prepare's per-user fast-path back-off (Shard::rhint)
    DBuf<uint16_t> d_win;                          // window cache: 16-bit entries (two per 32-bit entry when a user has >= 65536 ratings)
    DBuf<T> d_ms, d_c, d_mcsr, d_b;
    DBuf<double> d_objp;
    std::vector<Bin> bins;
    std::vector<Bin> sbins;                      // sweep classes (k_vsweep_all)
    std::vector<Bin> pbins;                      // prepare classes (k_prepare_all)
    std::vector<Bin> ubins;                      // U-step bins: an extra class, long users get workgroup clusters
    unsigned* bar_p = nullptr;                    // cluster arrival counters: live behind the 68 counters of d_counters (one memset)
    size_t bar_n = 0;
    DBuf<char> d_xch;
    DBuf<unsigned long long> d_rowcnt;            // per U-step class: rows of V gathered since the solver was created (count_rows)
    size_t xch_stride = 0;
    int max_clusters = 1;
    // ---- eval data (0 = train, 1 = test)
    struct EvalSet {
        int64_t nnz = 0;
        DBuf<int64_t> uptr;
        DBuf<int32_t> item;
        DBuf<double> val, gain, idcg, disc;
        DBuf<uint16_t> elvl;              // dense rank of the RAW rating inside the user (util.cpp:471 compares doubles)
        DBuf<int64_t> erunofs;
        DBuf<int32_t> erunstart;
        int max_raw_levels = 0;
        std::vector<Bin> bins;
        std::vector<int64_t> h_uptr;
        std::vector<double> h_val;        // only for a set with more than 64 raw levels per user (real-valued ratings)
        // at most 64 raw levels: the level tables (count and gain per (user, level), slots as run_start) -- and on the device either
        // the set's own level arrays or, for the training set, the solver's
        std::vector<int64_t> h_runofs;
        std::vector<int32_t> h_cnt;
        std::vector<double> h_lgain;
        const uint16_t* elvl_p = nullptr;
        const int64_t* erunofs_p = nullptr;
        const int32_t* erunstart_p = nullptr;
        int idcg_k = -1;
    } ev[2];
    DBuf<double> d_out4;
    // ---- factors and CG vectors (d2 x ld, nu x ld)
    DBuf<T> d_U, d_V, d_Vnew, d_g, d_delta, d_rr, d_p, d_Hp;
    CGState* d_cgp = nullptr;                     // the CG scalars live in d_scal[32..43): they come back with the objective's read-back
    DBuf<double> d_partA, d_partB, d_scal;       // reduction partials, small scalar block
    DBuf<unsigned long long> d_counters;
    DBuf<char> d_scratch;
    size_t scratch_stride = 0;
    int scratch_blocks = 0;
    int u_big_blocks = 0;                         // scratch slices the concurrent big U-step bins need together
    double* h_scal = nullptr;                     // pinned
    double* h_uobj = nullptr;                     // pinned: the U step's objective sums (its own buffer: read late by a pipelined loop)
    bool ustep_pending = false;                   // a U step is queued whose results have not been read yet
    double fin_obj = 0.0;                         // results of the U step that update_V finished on the loop's behalf
    int64_t fin_info[2] = {0, 0};
    bool fin_ready = false;
    bool device_join = false;                     // pipelined loop: the solver's stream waits for the lanes on the device
    CGState* h_cg = nullptr;                      // = h_scal + 32
    unsigned long long* h_counters = nullptr;     // pinned
    int ew_blocks = 1, ew_per_block = 1;          // elementwise decomposition over d2*ld
    bool have_sorted = false;
    bool state_of_rejected_V = false;             // the sorted state belongs to a V_new the line search did not accept (q5)
    double unorm2 = 0.0;                          // all-rank |U|^2 of the current U
    bool unorm_valid = false;

    ~Solver() override {
        if (st) (void)hipStreamSynchronize(st);
#ifdef PCR_RESORT_STAT
        {   // developer build: how often the nearly-sorted fast path of the per-user sorts was taken
            unsigned long long h[4];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_resort_stat), sizeof(h)) == hipSuccess && h[0] + h[1])
                fprintf(stderr, "resort-stat: fast path %llu sorts (%.1f %%), %.1f %% of the sorted ratings; full network %llu sorts\n", h[0],
                        100.0 * h[0] / (h[0] + h[1]), 100.0 * h[2] / (double)(h[2] + h[3]), h[1]);
        }
#endif
#ifdef PCR_PREP_PROF
        {   // developer build: per-phase shader clocks of thread 0 of every k_prepare workgroup, by class
            unsigned long long h[32];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prep_prof), sizeof(h)) == hipSuccess) {
                static const char* ph[] = {"load", "sort", "write", "windows", "loss"};
                static const char* cl[] = {"64", "256", "512", "512g"};
                for (int c = 0; c < 4; ++c) {
                    if (!h[c * 8 + 7]) continue;
                    double tot = 0; for (int i = 0; i < 5; ++i) tot += (double)h[c * 8 + i];
                    fprintf(stderr, "prep-prof %-5s wgs %8llu kclk/wg %8.1f :", cl[c], h[c * 8 + 7], tot / 1000.0 / h[c * 8 + 7]);
                    for (int i = 0; i < 5; ++i) fprintf(stderr, " %s %.1f%%", ph[i], 100.0 * h[c * 8 + i] / tot);
                    fprintf(stderr, "\n");
                }
            }
        }
#endif
        prof_resolve();
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (ar_st) { (void)hipStreamSynchronize(ar_st); (void)hipStreamDestroy(ar_st); }
        for (hipEvent_t e : ev_rng) (void)hipEventDestroy(e);
        if (ev_ar) (void)hipEventDestroy(ev_ar);
        if (comm) ncclCommDestroy(comm);
        p2p.reset();
        if (h_scal) (void)hipHostFree(h_scal);
        if (h_uobj) (void)hipHostFree(h_uobj);
        if (h_counters) (void)hipHostFree(h_counters);
        for (int i = 0; i < NSIDE; ++i) if (side[i]) (void)hipStreamDestroy(side[i]);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (int i = 0; i < MAXLANE; ++i) if (ev_lane[i]) (void)hipEventDestroy(ev_lane[i]);
        if (hi) (void)hipStreamDestroy(hi);
        for (hipStream_t x : hi_spare) (void)hipStreamDestroy(x);
        if (ev_hi) (void)hipEventDestroy(ev_hi);
        if (st) (void)hipStreamDestroy(st);
    }

    // ------------------------------------------------------------------------------ profiling
    struct ProfScope {
        Solver* s; ProfSlot* slot = nullptr; hipEvent_t a = nullptr, b = nullptr; hipStream_t q;
        ProfScope(Solver* s_, const std::string& name, hipStream_t q_ = nullptr, int64_t ratings = -1, int64_t users = -1)
            : s(s_), q(q_ ? q_ : s_->st) {
            if (!s->prof_on) return;
            ProfSlot* sl = &s->prof[name];
            sl->ratings = ratings; sl->users = users;
            // sampled: an event pair costs ~3 us of queue time.  Slots launched once per outer iteration (the U-step classes,
            // the prepares, the fork..join walls) are sampled at least every 4th launch, so that a 20-step run still
            // averages five of them; the per-CG-iteration kernels every prof_period-th
            const bool rare = name.compare(0, 5, "ustep") == 0 || name.compare(0, 5, "wall:") == 0 || name.compare(0, 7, "prepare") == 0;
            const int period = rare ? std::min(s->prof_period, 4) : s->prof_period;
            if ((sl->seen++ % period) != 0) return;
            slot = sl;
            a = s->ev_get(); b = s->ev_get();
            (void)hipEventRecord(a, q);
        }
        ~ProfScope() {
            if (!slot) return;
            (void)hipEventRecord(b, q);
            slot->pending.emplace_back(a, b);
            slot->n += 1;
        }
    };
    std::vector<hipEvent_t> ev_pool;      // timing events are recycled: creating one per launch costs more than the launch
    hipEvent_t ev_get() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void prof_prewarm(int n) override {
        while ((int)ev_pool.size() < n) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) break; ev_pool.push_back(e); }
    }
    int prof_resolve() override {
        for (auto& kv : prof) {
            for (auto& pr : kv.second.pending) {
                float ms = 0.f;
                (void)hipEventSynchronize(pr.second);
                if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.ms += ms;
                ev_pool.push_back(pr.first); ev_pool.push_back(pr.second);
            }
            kv.second.pending.clear();
        }
        return PCR_OK;
    }

    // does work on stream b wait for work on stream a (same hardware queue)?
    int shares_queue(hipStream_t a, hipStream_t b, long long ticks, hipEvent_t ev, bool* out) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b);
        HIPCHK(hipEventRecord(ev, b));
        const auto t0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(ev));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        HIPCHK(hipStreamSynchronize(a));
        *out = us > 150.0;                                         // the spin lasts 300 us
        return PCR_OK;
    }
    // Do the hardware queues of `busy` and `other` share a command-processor PIPE?  Queues on one pipe share workgroup dispatch:
    // a queue that merely holds a waiting or spinning kernel slows every kernel of the other (tools/ubench/gate_probe.hip: 27 ->
    // 62-68 us per dependent 65536-workgroup kernel; NOTES.md round 3).  Measured the same way here: a chain of empty
    // 65536-workgroup kernels on `busy`, alone (-> *base_us per kernel, once) and with a spinning one-wave kernel on `other`
    // (twice, the smaller figure counts: noise only adds).  gate_probe's mode 10 is this probe against a known pair: 14.9 -> 24.1 us
    // per kernel on a shared pipe, 14.9 -> 14.9 otherwise; in the solver's own layout the third side lane reads 14.9 -> 17.9.
    int shares_pipe(hipStream_t busy, hipStream_t other, long long spin_ticks, double* base_us, bool* out) {
        constexpr int N = 12;
        hipEvent_t e0 = ev_get(), e1 = ev_get();
        auto chain = [&](double* us) -> int {
            HIPCHK(hipEventRecord(e0, busy));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_nop, dim3(65536), dim3(64), 0, busy);
            HIPCHK(hipEventRecord(e1, busy));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            *us = 1e3 * ms / N;
            return PCR_OK;
        };
        if (*base_us <= 0.0) { double warm; RC(chain(&warm)); RC(chain(base_us)); }
        double with = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, other, spin_ticks);
            double w = 0.0;
            RC(chain(&w));
            HIPCHK(hipStreamSynchronize(other));
            with = std::min(with, w);
        }
        ev_pool.push_back(e0); ev_pool.push_back(e1);
        *out = with > 1.1 * *base_us;                              // (a queue on another pipe reproduces the base figure to 0.1 us)
        if (tune.debug) fprintf(stderr, "[pcr] pipe probe: %.1f us per kernel alone, %.1f with the other queue held -> %s\n", *base_us, with, *out ? "SAME pipe" : "separate pipes");
        return PCR_OK;
    }
    int pick_lanes() {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, prm.device) != hipSuccess || khz <= 0) khz = 100000;
        const long long ticks = (long long)khz * 300 / 1000;       // 300 us
        for (int i = 0; i < MAXLANE; ++i) HIPCHK(hipEventCreateWithFlags(&ev_lane[i], hipEventDisableTiming));
        lane[0] = st; nlane = 1;
        if (tune.lanes == 1) return PCR_OK;                                                      // pcr_tune("lanes", "1"): no concurrency
        bool dummy = false;
        auto ensure_side = [&](int c) -> int { if (!side[c]) HIPCHK(hipStreamCreateWithFlags(&side[c], hipStreamNonBlocking)); return PCR_OK; };
        RC(ensure_side(0));
        RC(shares_queue(st, side[0], ticks / 30, ev_lane[0], &dummy));                          // warm up: first launches are slow
        const int want = tune.lanes > 0 ? std::min(MAXLANE, tune.lanes) : 4;
        for (int c = 0; c < NSIDE && nlane < want; ++c) {
            bool clash = false;
            RC(ensure_side(c));
            for (int l = 0; l < nlane && !clash; ++l) RC(shares_queue(lane[l], side[c], ticks, ev_lane[0], &clash));
            if (!clash) lane[nlane++] = side[c];
            if (tune.debug) fprintf(stderr, "[pcr] side stream %d %s\n", c, clash ? "shares a queue with a lane" : "is a lane");
        }
        // Which lane shares the solver's stream's command-processor pipe?  Until round 4 that was arranged by the order in which
        // the streams are created (and held only while the host application created none of its own before the solver); now it is
        // measured: a lane on the solver's pipe goes LAST (its class is throttled by, and throttles, the solver's stream: it takes
        // the class that is off the critical path), and the high-priority stream of the cluster class must NOT be on that pipe
        // (612 instead of 373 us for the cluster class when it is) -- up to three replacements are tried.
        const long long spin = ticks * 2;                          // 0.6 ms: longer than the probe's chain (12 x 15-24 us) even on a shared pipe
        double base_us = 0.0;
        std::vector<hipStream_t> keep, shared;
        for (int l = 1; l < nlane; ++l) {
            bool same = false;
            RC(shares_pipe(st, lane[l], spin, &base_us, &same));
            (same ? shared : keep).push_back(lane[l]);
        }
        int q = 1;
        for (hipStream_t x : keep) lane[q++] = x;
        for (hipStream_t x : shared) lane[q++] = x;
        pipe_lanes = (int)shared.size();
        for (int attempt = 0; attempt < 4; ++attempt) {
            bool same = false;
            RC(shares_pipe(st, hi, spin, &base_us, &same));
            hi_on_solver_pipe = same;
            if (!same || attempt == 3) break;
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            hipStream_t again = nullptr;
            HIPCHK(hipStreamCreateWithPriority(&again, hipStreamNonBlocking, greatest));
            hi_spare.push_back(hi);                                // (kept until the solver goes: destroying it would hand its queue to the next stream)
            hi = again;
        }
        if (tune.debug) fprintf(stderr, "[pcr] lanes: %d side lane(s), %d of them on the solver's pipe (placed last); the high-priority stream is %s the solver's pipe\n",
                                nlane - 1, pipe_lanes, hi_on_solver_pipe ? "ON" : "off");
        return PCR_OK;
    }

    // ------------------------------------------------------------------------------ setup
    // users by rating count, longest first, ties in user order (what a stable sort by descending length gives): a counting sort,
    // once per CSR -- every class layout below is then one pass over this list
    static void length_order(const std::vector<int64_t>& uptr, int64_t nu, std::vector<int32_t>& order) {
        order.resize((size_t)nu);
        int64_t maxlen = 0;
        for (int64_t u = 0; u < nu; ++u) maxlen = std::max(maxlen, uptr[u + 1] - uptr[u]);
        if (maxlen > ((int64_t)1 << 24)) {
            for (int64_t u = 0; u < nu; ++u) order[u] = (int32_t)u;
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return (uptr[a + 1] - uptr[a]) > (uptr[b + 1] - uptr[b]); });
            return;
        }
        std::vector<int64_t> start((size_t)maxlen + 2, 0);
        for (int64_t u = 0; u < nu; ++u) start[(size_t)(maxlen - (uptr[u + 1] - uptr[u])) + 1]++;
        for (int64_t l = 0; l <= maxlen; ++l) start[(size_t)l + 1] += start[(size_t)l];
        for (int64_t u = 0; u < nu; ++u) order[(size_t)start[(size_t)(maxlen - (uptr[u + 1] - uptr[u]))]++] = (int32_t)u;
    }
    static void make_bins(const std::vector<int64_t>& uptr, int64_t nu, const std::vector<int64_t>* runofs, std::vector<Bin>& out,
                          const std::vector<int32_t>& order,
                          const std::vector<int>& limits = {BIN_LIMIT[0], BIN_LIMIT[1], BIN_LIMIT[2]},
                          const std::vector<int>& blocks = {BIN_BLOCK[0], BIN_BLOCK[1], BIN_BLOCK[2], BIN_BLOCK[3]}) {
        const int nb = (int)limits.size() + 1;
        out.clear();
        out.resize(nb);
        for (int b = 0; b < nb; ++b) { out[b].block = blocks[b]; out[b].big = (b == nb - 1); out[b].limit = b < nb - 1 ? limits[b] : 0; }
        for (int64_t q = 0; q < nu; ++q) {       // longest first: the tail of a launch is made of short users
            const int32_t u = order[(size_t)q];
            int64_t len = uptr[u + 1] - uptr[u];
            int b = 0;
            while (b < nb - 1 && len > limits[b]) ++b;
            out[b].users.push_back(u);
            out[b].nnz += len;
            out[b].cap = std::max<int>(out[b].cap, (int)len);
            if (runofs) out[b].max_lev = std::max<int>(out[b].max_lev, (int)((*runofs)[u + 1] - (*runofs)[u]) - 1);
        }
    }

    // The SpMM plan: tiles / chunk lists / workgroup map on the host (pcr_plan.h), the tile-major CSC, flags and slab rows on the
    // device (pcr_plan_dev.h) from d_uptr / d_item.  Fills d_ruser, d_c2r, d_crow, d_cuser, d_cuf, d_chunk_ptr, d_slot_base,
    // d_slot_id, d_item_slot, d_blk_chunks and P's host-side fields.
    template <typename K>
    int build_plan_keys(const SpmmPlanIn& in, SpmmPlan& P, const std::vector<int64_t>& tile_u, std::vector<int64_t>& cut) {
        const int64_t nu = in.nu, n = in.nnz, ntiles = P.ntiles;
        const int n_rng_ = P.n_rng;
        RC(d_ruser.alloc((size_t)n)); RC(d_c2r.alloc((size_t)n)); RC(d_crow.alloc((size_t)n)); RC(d_cuser.alloc((size_t)n));
        cut.assign((size_t)ntiles * (n_rng_ + 1), 0);
        for (int64_t t = 0; t < ntiles; ++t) {                    // (one item range: the cuts are the tiles' own bounds)
            cut[(size_t)t * (n_rng_ + 1)] = in.uptr[tile_u[t]];
            cut[(size_t)t * (n_rng_ + 1) + n_rng_] = in.uptr[tile_u[t + 1]];
        }
        if (n == 0) return PCR_OK;
        DBuf<K> key, skey;
        DBuf<int32_t> val, sval;
        DBuf<int64_t> d_tile_u;
        RC(key.alloc((size_t)n)); RC(skey.alloc((size_t)n)); RC(val.alloc((size_t)n)); RC(sval.alloc((size_t)n));
        RC(d_tile_u.upload(tile_u, st));
        const unsigned grid_u = (unsigned)std::min<int64_t>(1 << 20, cdiv(std::max<int64_t>(nu, 1), 4));
        const bool excl = vblock_nbp > 0;
        hipLaunchKernelGGL((k_plan_keys<K>), dim3(grid_u), dim3(256), 0, st, d_uptr.p, d_item.p, d_tile_u.p, (int)ntiles, d2, nu, key.p, val.p, d_ruser.p,
                           excl ? d_excl.p : (const unsigned char*)nullptr);
        HIPCHK(hipGetLastError());
        const int bits = plan_bits((unsigned long long)(ntiles + (excl ? 1 : 0)) * (unsigned long long)std::max<int64_t>(d2, 1));
        HIPCHK(plan_sort_pairs<K>(key.p, skey.p, val.p, sval.p, (size_t)n, bits, st));
        const unsigned grid_z = (unsigned)std::min<int64_t>(1 << 20, cdiv(n, 256));
        hipLaunchKernelGGL(k_plan_gather, dim3(grid_z), dim3(256), 0, st, sval.p, d_item.p, d_ruser.p, d_c2r.p, d_crow.p, d_cuser.p, n);
        HIPCHK(hipGetLastError());
        if (n_rng_ > 1 || excl) {                                 // where every tile's sorted entries begin and cross into each item range
            std::vector<K> probe;
            for (int64_t t = 0; t <= ntiles; ++t)
                for (int r = 0; r < (t < ntiles ? n_rng_ : 1); ++r) probe.push_back((K)t * (K)d2 + (K)P.rng_item[r]);
            DBuf<K> d_probe;
            DBuf<int64_t> d_pos;
            RC(d_probe.upload(probe, st)); RC(d_pos.alloc(probe.size()));
            hipLaunchKernelGGL((k_plan_lower_bounds<K>), dim3((unsigned)cdiv((int64_t)probe.size(), 256)), dim3(256), 0, st, skey.p, n, d_probe.p, (int)probe.size(), d_pos.p);
            HIPCHK(hipGetLastError());
            std::vector<int64_t> pos(probe.size());
            HIPCHK(hipMemcpyAsync(pos.data(), d_pos.p, pos.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int64_t t = 0; t < ntiles; ++t) {
                for (int r = 0; r < n_rng_; ++r) cut[(size_t)t * (n_rng_ + 1) + r] = pos[(size_t)t * n_rng_ + r];
                cut[(size_t)t * (n_rng_ + 1) + n_rng_] = pos[(size_t)(t + 1) * n_rng_];      // (the next tile's first entry; the excluded ratings' first for the last tile)
            }
        }
        HIPCHK(hipStreamSynchronize(st));                         // (the temporaries go out of scope)
        return PCR_OK;
    }
    int build_plan(const SpmmPlanIn& in, SpmmPlan& P) {
        std::vector<int64_t> tile_u, cut;
        plan_tiles(in, P, tile_u);
        const int64_t n = in.nnz;
        const unsigned long long key_span = (unsigned long long)(P.ntiles + (vblock_nbp ? 1 : 0)) * (unsigned long long)std::max<int64_t>(d2, 1);
        if (key_span < ((unsigned long long)1 << 32) && !tune.plan_key64) RC(build_plan_keys<uint32_t>(in, P, tile_u, cut));
        else RC(build_plan_keys<unsigned long long>(in, P, tile_u, cut));
        std::vector<int32_t> trc0;
        plan_chunks(P, cut.empty() ? n : cut.back(), cut, trc0);       // (cut.back(): the entries the tiles hold -- all of them unless a block is left out)
        plan_blocks(P, geo.G, trc0);
        const int64_t nchunks = (int64_t)P.chunk_ptr.size() - 1;
        RC(d_chunk_ptr.upload(P.chunk_ptr, st)); RC(d_blk_chunks.upload(P.blk, st));
        RC(d_cuf.alloc((size_t)n)); RC(d_slot_base.alloc((size_t)nchunks + 1)); RC(d_item_slot.alloc((size_t)d2 + 1));
        // new-item flags and the incidence count of every chunk; their prefix sums on the host (nchunks integers)
        std::vector<int32_t> inc_base((size_t)nchunks + 1, 0);
        const unsigned grid_c = (unsigned)std::min<int64_t>(1 << 20, cdiv(std::max<int64_t>(nchunks, 1), 4));
        if (nchunks > 0) {
            DBuf<int32_t> d_cnt;
            RC(d_cnt.alloc((size_t)nchunks));
            hipLaunchKernelGGL(k_plan_flags, dim3(grid_c), dim3(256), 0, st, d_chunk_ptr.p, nchunks, d_crow.p, d_cuser.p, d_cuf.p, d_cnt.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(inc_base.data() + 1, d_cnt.p, (size_t)nchunks * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int64_t c = 0; c < nchunks; ++c) inc_base[(size_t)c + 1] += inc_base[(size_t)c];
        }
        HIPCHK(hipMemcpyAsync(d_slot_base.p, inc_base.data(), inc_base.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        const int64_t n_inc = inc_base[(size_t)nchunks];
        P.slab_rows = (size_t)n_inc;
        RC(d_slot_id.alloc((size_t)n_inc));
        DBuf<int32_t> inc_item, inc_idx, sitem, sidx;
        RC(inc_item.alloc((size_t)n_inc)); RC(inc_idx.alloc((size_t)n_inc)); RC(sitem.alloc((size_t)n_inc)); RC(sidx.alloc((size_t)n_inc));
        if (n_inc > 0) {
            hipLaunchKernelGGL(k_plan_inc_items, dim3(grid_c), dim3(256), 0, st, d_chunk_ptr.p, nchunks, d_crow.p, d_slot_base.p, inc_item.p, inc_idx.p);
            HIPCHK(hipGetLastError());
            // the rows of one item are consecutive in the slab, in chunk order: a STABLE sort of the incidences by item
            HIPCHK(plan_sort_pairs<uint32_t>(reinterpret_cast<const uint32_t*>(inc_item.p), reinterpret_cast<uint32_t*>(sitem.p), inc_idx.p, sidx.p, (size_t)n_inc,
                                             plan_bits((unsigned long long)std::max<int64_t>(d2, 1)), st));
            hipLaunchKernelGGL(k_plan_slots, dim3((unsigned)std::min<int64_t>(1 << 20, cdiv(n_inc, 256))), dim3(256), 0, st, sidx.p, d_slot_id.p, n_inc);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_plan_item_slot, dim3((unsigned)std::min<int64_t>(1 << 20, cdiv(d2 + 1, 256))), dim3(256), 0, st, sitem.p, n_inc, d2, d_item_slot.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
#ifdef PCR_PLAN_CHECK
        RC(check_plan_against_host(in, P, inc_base));
#endif
        return PCR_OK;
    }
#ifdef PCR_PLAN_CHECK
    template <typename X>
    static bool same_as(const DBuf<X>& d, const std::vector<X>& h, const char* name) {
        std::vector<X> g(h.size());
        if (d.n < h.size()) { fprintf(stderr, "[plan check] %s: device %zu elements, host %zu\n", name, d.n, h.size()); return false; }
        if (!h.empty() && hipMemcpy(g.data(), d.p, h.size() * sizeof(X), hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (size_t i = 0; i < h.size(); ++i)
            if (memcmp(&g[i], &h[i], sizeof(X)) != 0) { fprintf(stderr, "[plan check] %s differs at %zu of %zu\n", name, i, h.size()); return false; }
        return true;
    }
    int check_plan_against_host(const SpmmPlanIn& in, const SpmmPlan& P, const std::vector<int32_t>& inc_base) {
        SpmmPlan H;
        build_spmm_plan_host(in, H);
        std::vector<int32_t> c2r((size_t)in.nnz);
        for (int64_t z = 0; z < in.nnz; ++z) c2r[(size_t)H.cpos[(size_t)z]] = (int32_t)z;
        bool ok = H.chunk == P.chunk && H.ntiles == P.ntiles && H.n_rng == P.n_rng && H.blocks == P.blocks && H.slab_rows == P.slab_rows &&
                  H.rng_item == P.rng_item && H.rng_blk == P.rng_blk && H.chunk_ptr == P.chunk_ptr && H.inc_base == inc_base &&
                  H.blk.size() == P.blk.size() && (H.blk.empty() || memcmp(H.blk.data(), P.blk.data(), H.blk.size() * sizeof(int2)) == 0);
        if (!ok) fprintf(stderr, "[plan check] host-side fields differ (chunk %d/%d tiles %d/%d blocks %d/%d slab %zu/%zu)\n", H.chunk, P.chunk, H.ntiles,
                         P.ntiles, H.blocks, P.blocks, H.slab_rows, P.slab_rows);
        ok = same_as(d_ruser, H.ruser, "ruser") && ok; ok = same_as(d_c2r, c2r, "c2r") && ok; ok = same_as(d_crow, H.crow, "crow") && ok;
        ok = same_as(d_cuser, H.cuser, "cuser") && ok; ok = same_as(d_cuf, H.cuf, "cuf") && ok; ok = same_as(d_slot_id, H.slot_id, "slot_id") && ok;
        ok = same_as(d_item_slot, H.item_slot, "item_slot") && ok;
        fprintf(stderr, "[plan check] %lld ratings, %d tiles, %d ranges, %zu slab rows: %s\n", (long long)in.nnz, P.ntiles, P.n_rng, P.slab_rows,
                ok ? "device plan == host plan" : "MISMATCH");
        if (!ok) { pcr_set_error("PCR_PLAN_CHECK: the device-built SpMM plan differs from the host-built one"); return PCR_ERR_STATE; }
        return PCR_OK;
    }
#endif

    // shard_first >= 0: `ds` holds ONLY this rank's users (renumbered from 0) -- users [shard_first, shard_first + ds.d1) of a
    // job with d1_total users (pcr_solver_create_shard); else the whole data set, partitioned here by pcr_partition_users
    int init(const pcr_dataset* ds, const pcr_params* p, int rank_, int nranks_, int64_t shard_first = -1, int64_t d1_total = 0) {
        prm = *p; rank = rank_; nranks = nranks_;
        const auto t_init = std::chrono::steady_clock::now();
        tune.read();
        if (prm.cg_max_iter == 0) prm.cg_max_iter = 10;      // zero-filled extension fields = the reference's constants
        if (prm.cg_tol == 0.0) prm.cg_tol = 0.01;
        if (prm.cg_max_iter < 0 || prm.cg_max_iter > 100000 || !(prm.cg_tol > 0.0)) { pcr_set_error("cg_max_iter / cg_tol out of range"); return PCR_ERR_ARG; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            pcr_set_error("no HIP device available: libprimalcr has no CPU fallback for the training path");
            return PCR_ERR_DEVICE;
        }
        if (prm.device < 0 || prm.device >= ndev) { pcr_set_error("device ordinal out of range"); return PCR_ERR_ARG; }
        HIPCHK(hipSetDevice(prm.device));
        {   // (one attribute, not hipGetDeviceProperties: that call fills a hundred fields, some of them through slow queries)
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, prm.device) != hipSuccess) cus = 0;
            ncu = cus > 0 ? cus : 256;
        }
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_hi, hipEventDisableTiming));
        {
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
        }
        // (the side streams are created when pick_lanes probes them: a solver that needs three lanes creates four or five streams,
        // not twelve -- a stream is 4-5 ms of set-up and a hardware queue of the device.  The first four, which every default
        // layout ends up creating, come up on a helper thread while this one prepares and uploads the shard: 20 ms of the set-up's
        // 80 on the ml1m shape, profiles/r06_cli_create.txt.)
        std::thread side_maker;
        struct JoinSide { std::thread& t; ~JoinSide() { if (t.joinable()) t.join(); } } join_side{side_maker};
        if (tune.lanes != 1) {
            const int want = tune.lanes > 0 ? std::min(MAXLANE, tune.lanes) : 4;
            side_maker = std::thread([this, dev = prm.device, n = std::min(NSIDE, want)]() {
                if (hipSetDevice(dev) != hipSuccess) return;
                for (int c = 0; c < n; ++c)
                    if (hipStreamCreateWithFlags(&side[c], hipStreamNonBlocking) != hipSuccess) { side[c] = nullptr; (void)hipGetLastError(); return; }
            });
        }

        const PcrCsr& X = ds->train;
        d1 = X.d1; d2 = X.d2; tnnz_file = ds->tnnz_file;
        if (prm.k < 1) { pcr_set_error("rank k must be >= 1"); return PCR_ERR_ARG; }
        if (d2 >= (int64_t)1 << 31 || d1 >= (int64_t)1 << 31) { pcr_set_error("d1/d2 must fit in int32"); return PCR_ERR_UNSUPPORTED; }
        geo.r = prm.k;
        geo.ld = (prm.k + 3) & ~3;
        geo.nchunk = geo.ld / VecOf<T>::N;
        geo.G = std::min(64, host_pow2(geo.nchunk));

        int64_t ds_u0 = 0;                                   // this rank's first user inside the data set's arrays
        if (shard_first >= 0) {
            if (d1_total < shard_first + X.d1 || d1_total >= (int64_t)1 << 31) { pcr_set_error("pcr_solver_create_shard: the shard does not fit the job's user range"); return PCR_ERR_ARG; }
            if (ds->test.d1 != X.d1) { pcr_set_error("pcr_solver_create_shard: train and test user counts differ"); return PCR_ERR_ARG; }
            first_user = shard_first; n_users = X.d1; d1 = d1_total;
        } else {
            std::vector<int64_t> bounds(nranks + 1);
            RC(pcr_partition_users(X.index.data(), d1, nranks, bounds.data()));
            first_user = bounds[rank];
            n_users = bounds[rank + 1] - bounds[rank];
            ds_u0 = first_user;
        }
        const int64_t z0 = X.index[ds_u0], z1 = X.index[ds_u0 + n_users];
        nnz_local = z1 - z0;
        if (nnz_local >= ((int64_t)1 << 31) - 1) { pcr_set_error("more than 2^31 ratings on one GPU"); return PCR_ERR_UNSUPPORTED; }
        const int64_t nu = n_users;

        // (pcr_tune("debug"): wall time of the set-up phases)
        // wall time of the set-up phases: kept (pcr_solver_setup_phase; omp-pmf-train --timing prints them), printed with pcr_tune("debug")
        auto t_phase = t_init;
        auto phase = [&](const char* what) {
            const auto now = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(now - t_phase).count();
            setup_ms.emplace_back(what, ms);
            if (tune.debug) fprintf(stderr, "[pcr] set-up: %-28s %8.1f ms\n", what, ms);
            t_phase = now;
        };
        phase("device, streams, events");
        // ---- host-side shard preparation
        std::vector<int64_t> uptr(nu + 1);
        for (int64_t u = 0; u <= nu; ++u) uptr[u] = X.index[ds_u0 + u] - z0;
        const int32_t* item = X.item.data() + z0;            // (the data set's own array: uploaded from where it lies)
        PcrLevels lv;
        std::string err;
        phase("copy CSR");
        int rc = pcr_build_levels(X, ds_u0, ds_u0 + nu, prm.solver_type, lv, err);
        if (rc != PCR_OK) { pcr_set_error(err); return rc; }
        phase("levels");
        // the shard's CSR goes up first: the nnz-sized part of the SpMM plan is built from it on the device (pcr_plan_dev.h)
        RC(d_uptr.upload(uptr, st)); RC(d_item.upload_n(item, (size_t)nnz_local));
        phase("CSR upload");
        std::vector<int32_t> by_len;
        length_order(uptr, nu, by_len);
        if (tune.vblock_users > 0 && nu > 0) {
            // The blocked-user V step (pcr_vblock.h): the users with the most ratings -- at most vblock_users of them, those that rate
            // at least a sixteenth of the catalogue -- go through dense MFMA kernels; the sparse plan leaves their ratings out.
            // The dense (user, item) -> CSR position table has ONE slot per pair, and the loader keeps a (user, item) pair that the
            // rating file holds twice as two ratings (as the reference's convert() does): a user with such a pair stays in the
            // sparse plan.  The table is capped at 1 GiB (a block of 256 users at the Yahoo!Music shape's d2 would be 140 MB).
            const int64_t blk_cap = std::min<int64_t>(tune.vblock_users, (((int64_t)1 << 30) / 4 / std::max<int64_t>(d2, 1)) & ~(int64_t)31);
            std::vector<int32_t> blk;
            std::vector<int32_t> cpos;
            for (int64_t q = 0; q < nu && (int64_t)blk.size() < blk_cap; ++q) {
                const int32_t u = by_len[(size_t)q];
                if ((uptr[u + 1] - uptr[u]) * 16 < d2) break;
                const size_t row = blk.size() * (size_t)d2;
                cpos.resize(row + (size_t)d2, -1);
                bool dup = false;
                for (int64_t z = uptr[u]; z < uptr[u + 1] && !dup; ++z) {
                    int32_t& slot = cpos[row + (size_t)item[z]];
                    dup = slot != -1;
                    slot = (int32_t)z;
                }
                if (dup) {
                    cpos.resize(row);
                    if (tune.debug) fprintf(stderr, "[pcr] blocked-user V step: user %d rates an item twice -- left in the sparse plan\n", (int)u);
                    continue;
                }
                blk.push_back(u);
            }
            if (!blk.empty()) {
                vblock_nbp = ((int)blk.size() + 31) & ~31;
                std::vector<unsigned char> excl((size_t)nu, 0);
                cpos.resize((size_t)vblock_nbp * (size_t)d2, -1);
                for (int32_t u : blk) excl[(size_t)u] = 1;
                blk.resize((size_t)vblock_nbp, -1);
                RC(d_blk_user.upload(blk, st)); RC(d_cpos_dense.upload(cpos, st)); RC(d_excl.upload(excl, st));
                if (tune.debug) fprintf(stderr, "[pcr] blocked-user V step: %d users (padded to %d) x %lld items through the dense kernels\n",
                                        (int)std::count(excl.begin(), excl.end(), 1), vblock_nbp, (long long)d2);
            }
        }
        SpmmPlan P;
        // (a block left out of the sparse plan: one item range -- the dense kernels add their share before the one exchange)
        const SpmmPlanIn plan_in{uptr, item, nu, nnz_local, d2, geo.ld, geo.G, ncu, sizeof(T), tune.spmm_chunk, tune.spmm_tiles, vblock_nbp ? 0 : tune.allreduce_chunks};
        RC(build_plan(plan_in, P));
        spmm_chunk = P.chunk; n_rng = P.n_rng; rng_item = P.rng_item; rng_blk = P.rng_blk; spmm_blocks = P.blocks; spmm_tiles = P.ntiles;
        if (n_rng > 1) {          // the all-reduce of a finished item range runs on its own stream (launch_spmm)
            HIPCHK(hipStreamCreateWithFlags(&ar_st, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev_ar, hipEventDisableTiming));
            ev_rng.resize(n_rng);
            for (auto& e : ev_rng) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        RC(d_slab.alloc(std::max<size_t>(P.slab_rows, 1) * geo.ld));
        sddmm_csc = (size_t)d2 * geo.ld * sizeof(T) > ((size_t)32 << 20) && spmm_tiles >= 8;     // item table larger than all L2s together
        if (tune.sddmm_csc >= 0) sddmm_csc = tune.sddmm_csc != 0;
        if (vblock_nbp) sddmm_csc = true;                          // the CG's SDDMM walks the plan (which leaves the block out), b in CSR order
        phase("tile-major CSC, slab plan");
        make_bins(uptr, nu, &lv.run_ofs, bins, by_len);
        for (auto& b : bins) RC(b.d_users.upload(b.users, st));
        // sweep / prepare classes: class 0 = one wave per user, class 1 = one 512-thread workgroup, class 2 = global scratch.
        // The sweeps keep 12 B per rating in LDS.  Where to cut between "a wave per user, eight users per workgroup" and "a
        // workgroup per user": a higher cut turns whole workgroups into waves (fewer workgroups to run through the CUs) but
        // lengthens the one-wave chains and, past 384, the LDS of eight waves leaves 3 instead of 4 workgroups per CU.
        // Cost model = rounds of workgroups through the chip x (1 + cut / 1024), over the candidate cuts (measured: ml1m 256:
        // 20.4-22.7 us per sweep, 320: 19.4, 384: 20.1-21.0, 512: 25.9; 10 M-rating Netflix-shaped slice 256: 181, 512: 163).
        int sweep_wave_cap = 256;
        {
            std::vector<int64_t> lens(nu);               // ascending
            for (int64_t q = 0; q < nu; ++q) { const int32_t u = by_len[(size_t)(nu - 1 - q)]; lens[q] = uptr[u + 1] - uptr[u]; }
            const int64_t max_lds = std::upper_bound(lens.begin(), lens.end(), (int64_t)4096) - lens.begin();     // users that fit LDS
            const int64_t cap_b = max_lds > 0 ? lens[max_lds - 1] : 0;
            double best = 0.0;
            for (int c : {256, 320, 384, 448, 512}) {
                const int64_t n_wave = std::upper_bound(lens.begin(), lens.end(), (int64_t)c) - lens.begin();
                const int64_t n_blk = std::max<int64_t>(0, max_lds - n_wave);
                const size_t wave_lds = 8 * ((size_t)c * sizeof(T) + (size_t)(c + 1) * 8 + 64);
                const size_t blk_lds = n_blk > 0 ? (size_t)cap_b * sizeof(T) + (size_t)(cap_b + 1) * 8 + 1024 : 0;
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, ((size_t)160 << 10) / std::max<size_t>(1, std::max(wave_lds, blk_lds))));
                const double rounds = (double)(cdiv(n_wave, 8) + n_blk) / ((double)ncu * per_cu);
                const double cost = std::max(rounds, 1.0) * (1.0 + c / 1024.0);
                if (best == 0.0 || cost < best) { best = cost; sweep_wave_cap = c; }
            }
        }
        if (tune.debug) fprintf(stderr, "[pcr] sweep wave cap %d\n", sweep_wave_cap);
        make_bins(uptr, nu, &lv.run_ofs, sbins, by_len, {std::min(sweep_wave_cap, 4095), 4096}, {64, 512, 512});
        for (auto& b : sbins) RC(b.d_users.upload(b.users, st));
        const int prep_wave_cap = 256;
        make_bins(uptr, nu, &lv.run_ofs, pbins, by_len, {prep_wave_cap, 4096}, {64, 512, 512});
        for (auto& b : pbins) RC(b.d_users.upload(b.users, st));
        // (measured, ml1m: 1024-thread teams make both launches slower -- k_vsweep_all 23 -> 31 us, k_prepare_all 99 -> 133 us:
        // sixteen one-wave users per workgroup cost more occupancy than the longest user's chain gains -- so 512 stays)
        sweep_pf4 = true;
        // U step: users with more than 1024 ratings are bound by one CU's gather bandwidth -> clusters of 4 workgroups
        // The U step keeps each user's rows of V in LDS (k_ustep, stage_rows), so its occupancy is set by LDS bytes, not
        // registers: finer length classes than the V side, and a workgroup size that grows with the class.
        // pcr_tune("ubins", "cap:block:resident,...") overrides the classes below 1024.
        // Measured (ml1m shape, k = 100): residency pays for users of <= 32 ratings (13 KB of rows: 10 one-wave workgroups
        // per CU still fit); above that the LDS image costs more occupancy than the faster passes gain -- the 33..64 class
        // was resident until its non-resident form got leaner (108 VGPRs against 124): 1.622 -> 1.604 ms, 10 M-rating
        // Netflix-shaped slice 82.5 -> 80.9 ms over 4 iterations; 65..128 resident: 1.80 ms -- so those classes gather from
        // the L2s with 16 waves per CU.
        std::vector<int> ucap = {32, 64, 128, 512}, ublk = {64, 64, 64, 256}, ures = {1, 0, 0, 0};
        if (const char* e = tune.ubins.empty() ? nullptr : tune.ubins.c_str()) {                 // "cap:block:resident,..."
            ucap.clear(); ublk.clear(); ures.clear();
            for (const char* q = e; *q;) {
                int c = 0, bl = 0, rs = 1, used = 0;
                if (sscanf(q, "%d:%d:%d%n", &c, &bl, &rs, &used) != 3 || (bl != 64 && bl != 256) || c < 1 || c >= 1024 || (rs && bl != 64) ||
                    (!ucap.empty() && c <= ucap.back())) { pcr_set_error("bad pcr_tune ubins"); return PCR_ERR_ARG; }
                ucap.push_back(c); ublk.push_back(bl); ures.push_back(rs);
                q += used; if (*q == ',') ++q;
            }
        }
        // Dual form (pcr_gram.h): users with at most gram_cap ratings run k_ustep_gram -- one class in place of the one-wave
        // classes.  gram_cap = pcr_tune("ustep_gram") or the largest count whose LDS (row image / Gram matrix + n-vectors)
        // still lets two workgroups share a CU, at most 128 (fp64: 64).
        int gram_cap = 0;
        if (tune.ustep_gram != 0 && tune.ubins.empty()) {
            const int hard = sizeof(T) == 4 ? 128 : 64;
            const int want = tune.ustep_gram > 0 ? std::min(tune.ustep_gram, hard) : GRAM_DEFAULT_CAP;
            const int nchp0 = geo.nchunk | 1;
            for (int c = want; c >= 16; c -= 8)
                if (gram_bytes<T>(c, host_pow2(c), lv.max_levels + 2, geo.ld, nchp0, 256) <= (tune.ustep_gram > 0 ? (size_t)160 : (size_t)80) * 1024) { gram_cap = c; break; }
            if (lv.max_levels > 64) gram_cap = 0;              // (real-valued ratings under PrimalCR: a level per rating -- keep the general kernel)
        }
        size_t ngram = 0;
        if (gram_cap > 64) { ucap = {64, gram_cap, 512}; ublk = {64, 256, 256}; ures = {0, 0, 0}; ngram = 2; }       // one wave up to 64 ratings
        else if (gram_cap > 0) { ucap = {gram_cap, 512}; ublk = {64, 256}; ures = {0, 0}; ngram = 1; }
        const size_t nsmall = ucap.size();
        // Latency or throughput?  A class with few users is one round of workgroups and is bound by the per-user dependency
        // chain: 512 threads, 8 rows in flight per lane group, 174-205 VGPRs = one workgroup per CU.  A class with many users
        // is bound by how busy each CU's memory pipe stays: smaller / leaner workgroups, so that two share a CU and one
        // gathers while the other scans or sorts (<= 1024 ratings: 256 threads; above: 512 threads at 4 rows in flight =
        // 124 VGPRs, and a class boundary at 2048 so that the per-rating arrays of two fit the LDS).  "Many" is more than
        // CUs/4 users: the greedy one-per-CU workgroups of all long classes together must leave CUs for the short classes
        // (ml1m: 88 + 221 users in throughput form 2.09 -> 2.03 ms per iteration; Netflix shape: U step 87 -> 69 ms).
        const int force_mode = tune.ustep_mode;                                                  // 1 latency, 2 throughput
        const int64_t many_users = std::max<int64_t>(1, ncu / 4);
        auto many = [&](int64_t users) { return force_mode ? force_mode == 2 : users > many_users; };
        int64_t n_mid = 0;
        for (int64_t u = 0; u < nu; ++u) { const int64_t len = uptr[u + 1] - uptr[u]; n_mid += len > 1024 && len <= 4096; }
        ucap.push_back(1024); ublk.push_back(512);
        if (many(n_mid)) { ucap.push_back(2048); ublk.push_back(512); }
        ucap.push_back(4096); ublk.push_back(512); ublk.push_back(512);
        make_bins(uptr, nu, &lv.run_ofs, ubins, by_len, ucap, ublk);
        for (size_t q = 0; q < ngram; ++q) ubins[q].gram = true;
        // Workgroup clusters trade throughput for latency: only the longest users of the shard (the critical path, more than
        // 1024 ratings) get them, ncu/(4K) users (all their workgroups fit the chip at once, see below) -- ONE extra class
        // whatever length class they came from (in global scratch if any of them needs it).  pcr_tune("cluster_k", "1") disables.
        int cluster_k = 4;
        if (tune.cluster_k != 4) cluster_k = 1;
        max_clusters = std::max(1, ncu / 2);
        if (cluster_k > 1) {
            Bin head;
            head.block = 512; head.K = cluster_k;
            // ncu / (4K) users = a quarter of the CUs: every cluster workgroup keeps a CU to itself (its LDS image) for the
            // whole launch, CUs the many short users cannot use meanwhile -- ml1m: 8 users 1.610 ms, 12-20: 1.59-1.61, 24: 1.63,
            // 32: 1.645, 48: 1.79 per iteration; 10 M-rating Netflix-shaped slice: U step 7.24 (32) -> 6.98 ms (16)
            size_t budget = (size_t)std::max(1, ncu / (4 * cluster_k));
            if (tune.cluster_users > 0) budget = (size_t)std::max(1, std::min(tune.cluster_users, ncu / cluster_k));
            for (size_t q = ubins.size(); q-- > nsmall + 1 && budget > 0;) {       // longest class first; users are sorted longest first
                Bin& b = ubins[q];
                const size_t take = std::min(budget, b.users.size());
                if (take == 0) continue;
                budget -= take;
                head.big = head.big || b.big;
                head.max_lev = std::max(head.max_lev, b.max_lev);
                for (size_t i = 0; i < take; ++i) {
                    const int32_t u = b.users[i];
                    const int64_t len = uptr[u + 1] - uptr[u];
                    head.users.push_back(u); head.nnz += len; head.cap = std::max<int>(head.cap, (int)len);
                    b.nnz -= len;
                }
                b.users.erase(b.users.begin(), b.users.begin() + take);
                b.cap = b.users.empty() ? 0 : (int)(uptr[b.users[0] + 1] - uptr[b.users[0]]);
            }
            if (!head.users.empty()) ubins.push_back(std::move(head));
        }
        for (size_t q = nsmall; q < ubins.size(); ++q) {
            Bin& b = ubins[q];
            // (a class whose per-rating arrays fill more than half the LDS runs one workgroup per CU whatever its register
            // count: it keeps the 8-rows-in-flight form)
            const bool lds_bound = !b.big && ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 4) > 72 * 1024;
            if (b.K > 1 || lds_bound || !many((int64_t)b.users.size())) { b.unr = 8; continue; }
            b.unr = 4;
            // (re-checked in round 3 for a 513..1024 class of 200 users, one round of workgroups: 256 threads at 8 rows in flight
            // 1.49 -> 1.69 ms per ml1m step, 512 threads at 4 rows: no change)
            if (b.limit == 1024 && !b.big) b.block = 256;
        }
        // A class whose per-rating arrays + r-vectors do not fit the 160 KB of LDS (fp64 at wide ranks with users near 4096
        // ratings, or thousands of rating levels under PrimalCR) runs the global-scratch form of the kernel instead.
        for (auto& b : ubins) {
            if (b.big || b.users.empty() || b.gram) continue;
            const size_t fixed = ustep_small_bytes(geo.ld, b.block, sizeof(T)) + ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 4);
            if (fixed > (size_t)160 * 1024) { b.big = true; b.block = 512; }
        }
        // the one-wave and 256-thread classes keep 4 rows in flight per lane group (8 measured on ml1m: the one-wave classes alone
        // 1.58 -> 1.67 ms per step, the 256-thread classes too 1.89 ms; Netflix shape U step 52 -> 68 ms: the registers cost more
        // occupancy than the deeper gathers gain)
        for (size_t q = 0; q < nsmall && q < ubins.size(); ++q)
            if (!ubins[q].gram && !ubins[q].users.empty()) ubins[q].unr = 4;
        u_big_blocks = 0;
        for (auto& b : ubins) {
            const int nus = (int)b.users.size();
            b.ugrid = b.K > 1 ? std::min(nus, std::max(1, ncu / b.K)) * b.K : (b.big ? std::min(nus, 2 * ncu) : nus);
            if (b.big) { b.scratch_ofs = u_big_blocks; u_big_blocks += b.ugrid; }
        }
        // window cache (pcr_kernels.h, Shard::win): one slot per other level, up to 9 levels
        const int sh_ws_for_bins = (tune.window_cache && lv.max_levels >= 2 && lv.max_levels <= 9) ? lv.max_levels - 1 : 0;
        {   // LDS residency: what is left of the 160 KB after the r-vectors and the per-rating arrays, in rows of V
            const int nchp = geo.nchunk | 1;
            // (capping the image at 128 / 112 / 96 / 64 KB, so that workgroups of the short classes could share the CU, changes
            // nothing: ml1m 1.424-1.438 ms per step at every cap, Netflix-shaped U step 62.6-62.8 ms -- NOTES.md round 4)
            const size_t lim = 160 * 1024;
            for (size_t bi = 0; bi < ubins.size(); ++bi) {
                Bin& b = ubins[bi];
                if (b.users.empty() || b.gram) continue;
                // the LDS image pays where LDS is spare: the one-wave classes of <= 64 ratings, and the latency-bound
                // 512-thread classes (one workgroup per CU anyway), which keep as many rows as fit beside their arrays
                const int res_on = bi < nsmall ? (b.block == 64 ? ures[bi] : 0) : (b.unr == 8);
                // the window rows of the class's longest user in LDS (16 bit), where that still leaves the class its occupancy:
                // every class of at most 1024 ratings (8 KB), the one-workgroup-per-CU classes whatever their length
                b.wcap = 0;
                if (tune.ustep_win_lds && !b.big && sh_ws_for_bins > 0 && (b.cap <= 1024 || b.unr == 8) &&
                    ustep_small_bytes(geo.ld, b.block, sizeof(T)) + ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 4) +
                        carve_bytes((size_t)b.cap * sh_ws_for_bins, 2) <= lim - 8 * 1024)
                    b.wcap = b.cap * sh_ws_for_bins;
                const size_t fixed = ustep_small_bytes(geo.ld, b.block, sizeof(T)) + carve_bytes(b.wcap, 2) +
                                     (b.big ? 0 : ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 4));
                const int64_t room = fixed < lim ? (int64_t)((lim - fixed) / ((size_t)nchp * 16)) : 0;
                const int64_t want = (b.cap + b.K - 1) / b.K;                  // longest slice a member gathers
                b.rcap = res_on ? (int)std::max<int64_t>(0, std::min(room, want)) : 0;
            }
        }
        for (auto& b : ubins) RC(b.d_users.upload(b.users, st));
        if (tune.ustep_newton && geo.ld <= NEWTON_MAX_LD) {
            // the exact-Newton mode (pcr_newton.h): users of 1 .. NEWTON_MAX_N ratings get their direction from the explicit Hessian
            std::vector<int32_t> nus_list;
            for (int64_t q = 0; q < nu; ++q) {
                const int32_t u = by_len[(size_t)q];
                const int64_t len = uptr[u + 1] - uptr[u];
                if (len >= 1 && len <= NEWTON_MAX_N) { nus_list.push_back(u); newton_cap = std::max<int>(newton_cap, (int)len); }
            }
            newton_n = (int)nus_list.size();
            newton_rs = lv.max_levels + 2;
            RC(d_newton_users.upload(nus_list, st));
            RC(d_dir.alloc((size_t)std::max<int64_t>(nu, 1) * geo.ld));
            const int ldp = (geo.ld + 15) & ~15;
            newton_grid = std::max(1, std::min(newton_n, 2 * ncu));
            newton_stride = (((size_t)newton_cap + 1) * ldp * sizeof(double) + 255) & ~(size_t)255;
            RC(d_newton_scratch.alloc(newton_stride * (size_t)newton_grid));
#ifndef PCR_NO_OPTIONAL_KERNELS
            HIPCHK(hipFuncSetAttribute((const void*)k_unewton<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
            if (tune.debug) fprintf(stderr, "[pcr] exact-Newton U step: %d of %lld users through the explicit Hessian (<= %d ratings)\n", newton_n, (long long)nu, NEWTON_MAX_N);
        }
        {
            size_t need_x = 0;
            for (auto& b : ubins)
                if (b.K > 1 && !b.users.empty()) need_x = std::max(need_x, ustep_xch_bytes<T>(host_pow2(b.cap), geo.ld, b.K));
            xch_stride = (need_x + 255) & ~(size_t)255;
            bar_n = (size_t)max_clusters * ubins.size();               // the classes run concurrently: one set per class
            RC(d_xch.alloc(xch_stride * (size_t)max_clusters * ubins.size()));
            RC(d_rowcnt.alloc(ubins.size()));
            HIPCHK(hipMemset(d_rowcnt.p, 0, std::max<size_t>(ubins.size(), 1) * sizeof(unsigned long long)));
        }
        {   // two length classes that run the same workgroup form get kernel symbols of their own (k_ustep's CLS), so that
            // rocprofv3's per-symbol durations and PMC bytes belong to one class each
            std::map<std::string, int> seen;
            for (auto& b : ubins) {
                if (b.users.empty() || b.gram) continue;
                const std::string key = std::to_string(b.block) + (b.big ? "g" : "") + "k" + std::to_string(b.K) + (b.rcap > 0 ? "r" : "") + "u" + std::to_string(b.unr);
                const bool two = !b.big && b.K == 1 && b.rcap == 0 && b.unr == 4;       // the forms instantiated twice (set_lds_limits)
                b.sym = two ? (seen[key]++ & 1) : 0;
                // the 512-thread throughput form: symbol 0 is the register-capped one (two workgroups per CU), compiled for at most
                // half a CU's LDS; a class that needs more LDS than that takes symbol 1 (pcr_kernels.h, k_ustep)
                if (two && b.block == 512 && sizeof(T) == 4)
                    b.sym = ustep_small_bytes(geo.ld, b.block, sizeof(T)) + carve_bytes(b.wcap, 2) + ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 4) > (size_t)80 * 1024 ? 1 : 0;
            }
        }

        phase("length classes");
        RC(d_lvl.upload(lv.level, st));
        RC(d_runofs.upload(lv.run_ofs, st)); RC(d_runstart.upload(lv.run_start, st));
        RC(d_ms.alloc(nnz_local)); RC(d_sitem.alloc(nnz_local)); RC(d_slvl.alloc(nnz_local));
        RC(d_c.alloc(nnz_local)); RC(d_objp.alloc(nu)); RC(d_mcsr.alloc(nnz_local)); RC(d_b.alloc(nnz_local));
        sh.nu = nu; sh.nnz = nnz_local; sh.d2 = (int)d2;
        sh.uptr = d_uptr.p; sh.item = d_item.p; sh.lvl = d_lvl.p;
        sh.runofs = d_runofs.p; sh.runstart = d_runstart.p;
        sh.ms = d_ms.p; sh.sitem = d_sitem.p; sh.slvl = d_slvl.p; sh.objp = d_objp.p;
        RC(d_sidx.alloc(nnz_local)); RC(d_objr.alloc(nu));
        sh.sidx = d_sidx.p; sh.objr = d_objr.p;
        sh.ws = sh_ws_for_bins;
        sh.resort_d = std::max(0, std::min(tune.resort_window, 64));
        sh.prev_valid = 0;                                  // set once the first k_prepare of the solver's life has been queued
        RC(d_rhint.alloc(std::max<int64_t>(nu, 1)));
        HIPCHK(hipMemsetAsync(d_rhint.p, 0, std::max<int64_t>(nu, 1), st));
        sh.rhint = d_rhint.p;
        {
            int64_t longest = 0;
            for (int64_t u = 0; u < nu; ++u) longest = std::max(longest, uptr[u + 1] - uptr[u]);
            sh.w16 = (longest < 65536 && tune.win16) ? 1 : 0;
        }
        RC(d_win.alloc((size_t)nnz_local * sh.ws * (sh.w16 ? 1 : 2)));
        sh.win = d_win.p;

        phase("uploads, state arrays");
        // ---- eval sets (train shard, test shard)
        // The evaluator compares RAW ratings (util.cpp:471): per user the dense rank of the raw value (elvl + run tables), the gain
        // 2^v - 1 per rating (util.cpp:519) and, per ndcg_k, the ideal DCG.  With at most 64 raw levels per user (every rating
        // scale in use) all of it derives from the per-user LEVEL TABLES: the gain of a rating is the gain of its level --
        // pow() runs once per (user, level) on the host, the per-rating array is filled on the device -- and the ideal DCG walks
        // the level counts from the top; neither the 8-byte ratings nor a per-rating gain array cross PCIe.  The training set
        // reuses the solver's own level arrays when its raw levels are the rounded ones (integer ratings, or PrimalCR).
        for (int w = 0; w < 2; ++w) {
            const PcrCsr& E = w == 0 ? ds->train : ds->test;
            EvalSet& es = ev[w];
            const int64_t a = E.index[ds_u0], b = E.index[ds_u0 + nu];
            es.nnz = b - a;
            es.h_uptr.resize(nu + 1);
            for (int64_t u = 0; u <= nu; ++u) es.h_uptr[u] = E.index[ds_u0 + u] - a;
            if (w == 1) { RC(es.uptr.upload(es.h_uptr, st)); RC(es.item.upload_n(E.item.data() + a, (size_t)es.nnz)); }
            RC(es.idcg.alloc(nu));
            const bool same = w == 0 && (prm.solver_type == PCR_SOLVER_PCR || lv.integer_valued);
            PcrLevels rl_own;
            std::string e2;
            const PcrLevels* rl = same ? &lv : &rl_own;
            const bool have_levels = same || pcr_build_levels(E, ds_u0, ds_u0 + nu, PCR_SOLVER_PCR, rl_own, e2) == PCR_OK;
            std::vector<int32_t> order_own;
            if (w == 1) length_order(es.h_uptr, nu, order_own);
            const std::vector<int32_t>& order = w == 0 ? by_len : order_own;
            if (have_levels) {
                es.max_raw_levels = rl->max_levels;
                if (same) { es.elvl_p = d_lvl.p; es.erunofs_p = d_runofs.p; es.erunstart_p = d_runstart.p; }       // (uploaded with the shard)
                else {
                    RC(es.elvl.upload(rl->level, st)); RC(es.erunofs.upload(rl->run_ofs, st)); RC(es.erunstart.upload(rl->run_start, st));
                    es.elvl_p = es.elvl.p; es.erunofs_p = es.erunofs.p; es.erunstart_p = es.erunstart.p;
                }
                make_bins(es.h_uptr, nu, &rl->run_ofs, es.bins, order);
            } else {
                es.max_raw_levels = 1 << 30;
                make_bins(es.h_uptr, nu, nullptr, es.bins, order);
            }
            if (es.max_raw_levels <= 64) {
                // level tables: per (user, level) the count and the gain (pow(2, v) - 1 with the host's libm, as the reference computes it)
                es.h_runofs = rl->run_ofs;
                es.h_cnt.resize(rl->run_start.size());
                es.h_lgain.resize(rl->run_start.size());
                pcr_parallel_ranges(nu, pcr_host_threads(), [&](int, int64_t lo, int64_t hi) {
                    for (int64_t u = lo; u < hi; ++u) {
                        const int64_t o = rl->run_ofs[u], Tn = rl->run_ofs[u + 1] - o - 1;
                        for (int64_t l = 0; l < Tn; ++l) {
                            es.h_cnt[o + l] = rl->run_start[o + l + 1] - rl->run_start[o + l];
                            es.h_lgain[o + l] = pow(2.0, rl->lev_val[o + l]) - 1.0;
                        }
                        es.h_cnt[o + Tn] = 0; es.h_lgain[o + Tn] = 0.0;
                    }
                });
                DBuf<double> d_lgain;
                RC(d_lgain.upload(es.h_lgain, st));
                RC(es.gain.alloc((size_t)es.nnz));
                if (es.nnz > 0) {
                    const int64_t* up = w == 0 ? d_uptr.p : es.uptr.p;
                    hipLaunchKernelGGL(k_gain_from_levels, dim3((unsigned)std::min<int64_t>(65535 * 16, cdiv(nu, 4))), dim3(256), 0, st, up, es.elvl_p, es.erunofs_p,
                                       d_lgain.p, es.gain.p, nu);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipStreamSynchronize(st));              // (d_lgain goes out of scope)
                }
            } else {
                // more raw levels than the level-table form of the evaluator takes (real-valued ratings): the ratings themselves
                es.h_val.assign(E.val.begin() + a, E.val.begin() + b);
                RC(es.val.upload(es.h_val, st));
                std::vector<double> gain(es.nnz);
                pcr_parallel_ranges(es.nnz, pcr_host_threads(), [&](int, int64_t lo, int64_t hi) {
                    for (int64_t z = lo; z < hi; ++z) gain[z] = pow(2.0, es.h_val[z]) - 1.0;                  // util.cpp:519
                });
                RC(es.gain.upload(gain, st));
            }
            for (auto& bn : es.bins) RC(bn.d_users.upload(bn.users, st));
        }
        RC(d_out4.alloc(4 * (size_t)std::max<int64_t>(nu, 1)));

        phase("evaluation sets");
        // ---- factors / vectors
        const size_t nV = (size_t)d2 * geo.ld, nU = (size_t)nu * geo.ld;
        RC(d_U.alloc(nU)); RC(d_V.alloc(nV)); RC(d_Vnew.alloc(nV)); RC(d_g.alloc(nV)); RC(d_delta.alloc(nV));
        RC(d_rr.alloc(nV)); RC(d_p.alloc(nV)); RC(d_Hp.alloc(nV));
        HIPCHK(hipMemsetAsync(d_U.p, 0, std::max<size_t>(nU, 1) * sizeof(T), st));
        HIPCHK(hipMemsetAsync(d_V.p, 0, std::max<size_t>(nV, 1) * sizeof(T), st));
        ew_blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, cdiv((int64_t)nV, 1024)));   // 4 elements per thread: these kernels are latency-bound
        ew_per_block = cdiv((int64_t)nV, ew_blocks);
        RC(d_partA.alloc(4 * 2048)); RC(d_partB.alloc(4 * 2048)); RC(d_scal.alloc(64));
        RC(d_counters.alloc(4 + 64 + (bar_n + 1) / 2));
        bar_p = reinterpret_cast<unsigned*>(d_counters.p + 4 + 64);
        HIPCHK(hipHostMalloc((void**)&h_scal, 64 * sizeof(double)));
        HIPCHK(hipHostMalloc((void**)&h_uobj, 16 * sizeof(double)));
        static_assert(sizeof(CGState) <= 12 * sizeof(double), "CGState must fit d_scal[32..44)");
        h_cg = reinterpret_cast<CGState*>(h_scal + 32);
        d_cgp = reinterpret_cast<CGState*>(d_scal.p + 32);
        HIPCHK(hipHostMalloc((void**)&h_counters, (4 + 64) * sizeof(unsigned long long)));

        // ---- scratch for users that do not fit in LDS
        // (every class that can run the BIG form is sized with ITS OWN longest user and level count: the cluster class mixes
        // users of several length classes, and a mid-length user may hold more rating levels than any long one)
        size_t need = 0;
        size_t nbig = 0;
        for (const Bin* b : {&bins[3], &pbins[2]})
            if (!b->users.empty()) { need = std::max(need, prepare_bytes<T>(b->cap, host_pow2(b->cap), b->max_lev + 2, 8)); nbig = std::max(nbig, b->users.size()); }
        for (const Bin* b : {&bins[3], &sbins[2]})
            if (!b->users.empty()) { need = std::max(need, vsweep_bytes<T>(b->cap, b->max_lev + 2, true)); nbig = std::max(nbig, b->users.size()); }
        for (auto& b : ubins)
            if (b.big && !b.users.empty()) need = std::max(need, ustep_big_bytes<T>(b.cap, host_pow2(b.cap), b.max_lev + 2, 8));
        for (int w = 0; w < 2; ++w)
            if (!ev[w].bins[3].users.empty()) {
                need = std::max(need, std::max(eval_bytes<T>(ev[w].bins[3].cap), eval2_big_bytes<T>(ev[w].bins[3].cap, host_pow2(ev[w].bins[3].cap))));
                nbig = std::max(nbig, ev[w].bins[3].users.size());
            }
        if (need) {
            scratch_stride = (need + 255) & ~(size_t)255;
            scratch_blocks = std::max((int)std::min<size_t>(std::max<size_t>(nbig, 1), (size_t)ncu * 2), u_big_blocks);
            RC(d_scratch.alloc(scratch_stride * (size_t)scratch_blocks));
        }
        RC(set_lds_limits());
        HIPCHK(hipStreamSynchronize(st));
        phase("factors, scratch");
        if (side_maker.joinable()) side_maker.join();
        RC(pick_lanes());
        phase("stream lanes");
        return PCR_OK;
    }

    // opt in to > 64 KiB dynamic LDS for the 512-thread instantiations
    int set_lds_limits() {
        const int lim = 160 * 1024;
        HIPCHK(hipFuncSetAttribute((const void*)k_prepare<T, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_prepare_all<T, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep<T, 512, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep<T, 512, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep_all<T, false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep_all<T, true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep_all<T, false, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_vsweep_all<T, true, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
#define UL(BL, BG, KK, RS, UN) HIPCHK(hipFuncSetAttribute((const void*)k_ustep<T, BL, BG, KK, RS, UN>, hipFuncAttributeMaxDynamicSharedMemorySize, lim))
        UL(64, false, 1, true, 4); UL(64, false, 1, false, 4); UL(256, false, 1, false, 4);
#define UL1(BL) HIPCHK(hipFuncSetAttribute((const void*)k_ustep<T, BL, false, 1, false, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lim))
        UL1(64); UL1(256); UL1(512);
#undef UL1
        UL(512, false, 1, true, 8); UL(512, false, 4, true, 8);
        if (sizeof(T) == 4) HIPCHK(hipFuncSetAttribute((const void*)k_ustep<T, 512, false, 1, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        else UL(512, false, 1, false, 4);
        UL(512, true, 1, true, 8); UL(512, true, 1, false, 4); UL(512, true, 4, true, 8);
#undef UL
#ifndef PCR_NO_OPTIONAL_KERNELS      // (a measurement build without the three optional kernel families: profiles/r06_cli_create.txt)
        HIPCHK(hipFuncSetAttribute((const void*)k_ustep_gram<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_ustep_gram<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
#endif
        HIPCHK(hipFuncSetAttribute((const void*)k_eval<T, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHK(hipFuncSetAttribute((const void*)k_eval2<T, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        return PCR_OK;
    }

    // Launch one kernel per non-empty length bin, the bins concurrently: the largest bin stays on
    // the solver's stream, the others fork to side streams and join back (events, no host sync).
    template <class F>
    int for_bins(std::vector<Bin>& bs, const char* cls, F launch) {
        int main_bin = -1;
        size_t most = 0;
        for (size_t i = 0; i < bs.size(); ++i) if (bs[i].users.size() > most) { most = bs[i].users.size(); main_bin = (int)i; }
        if (main_bin < 0) return PCR_OK;
        // wall time of the whole concurrent group on the solver's stream (fork .. join): the per-bin slots
        // overlap, so their sum overstates the group's share of the timed region
        ProfScope wall(this, std::string("wall:") + cls, st);
        bool forked = false;
        // longest users first: their workgroups are the critical path and must not queue behind the many
        // short-user workgroups
        bool used[MAXLANE] = {};
        int next = 0;
        for (size_t ii = bs.size(); ii-- > 0;) {
            const size_t i = ii;
            if (bs[i].users.empty() || (int)i == main_bin) continue;
            const int l = nlane > 1 ? 1 + next++ % (nlane - 1) : 0;      // side lanes in turn; classes sharing a lane run back to back
            // cross-stream waits on events that are still pending are slow here (see update_V): drain the solver's
            // stream on the host before the fork, and join on the host too (join())
            if (l > 0 && !forked) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipEventRecord(ev_fork, st)); forked = true; }
            if (l > 0 && !used[l]) HIPCHK(hipStreamWaitEvent(lane[l], ev_fork, 0));
            used[l] = true;
            ProfScope ps(this, pname(cls, bs[i]), lane[l], bs[i].nnz, (int64_t)bs[i].users.size());
            launch(bs[i], lane[l]);
        }
        { ProfScope ps(this, pname(cls, bs[main_bin]), st, bs[main_bin].nnz, (int64_t)bs[main_bin].users.size()); launch(bs[main_bin], st); }
        for (int l = 1; l < nlane; ++l)
            if (used[l]) { HIPCHK(hipEventRecord(ev_lane[l], lane[l])); RC(join(ev_lane[l])); }
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    // the solver's stream continues after `ev`
    int join(hipEvent_t ev) {
        if (!device_join) HIPCHK(hipEventSynchronize(ev)); else HIPCHK(hipStreamWaitEvent(st, ev, 0));
        return PCR_OK;
    }
    // U step: the hardware runs only a few queues side by side (streams beyond that share a queue and serialise), so the
    // classes are placed deliberately: the cluster class (the longest users, the critical path) alone on a high-priority
    // stream; the other classes, longest first, dealt round-robin onto three streams, so the three longest start at once
    // and the short classes queue behind them and fill the CUs as those drain.
    template <class F>
    int for_ubins(F launch) {
        std::vector<Bin*> order;                       // [cluster class, then the others longest first]
        for (auto& b : ubins) if (!b.users.empty() && b.K > 1) order.push_back(&b);
        const size_t nhead = order.size();
        for (auto& b : ubins) if (!b.users.empty() && b.K <= 1) order.push_back(&b);
        if (order.empty()) return PCR_OK;
        std::stable_sort(order.begin() + nhead, order.end(), [](const Bin* a, const Bin* b) { return a->cap > b->cap; });
        // plan: (class, stream) in launch order.  Streams: 0..MAXLANE-1 = lane[], MAXLANE = hi.
        std::vector<std::pair<int, int>> plan;
        // (A longest-processing-time-first plan from class durations measured in the first U steps was tried: 2.10 ms
        // against 2.01 ms per iteration for this one -- durations measured side by side mislead it.)
        {
            for (size_t i = 0; i < nhead; ++i) plan.push_back({(int)i, MAXLANE});
            // round-robin over the lanes, longest class first -- with the first two lanes swapped: the solver's stream takes
            // the SECOND class (on the headline shape the many-user 256-thread class that finishes last, and the shortest
            // class behind it), so the join at the end finds the other lanes' events already signalled (1.803 -> 1.785 ms)
            // The classes beyond one per lane (the TAILS: short users, tens of microseconds) go behind the lanes in turn again, but
            // behind the cluster class before the third lane, and behind the solver's stream last: on the headline shape its
            // many-user class is the one that ends last (1.478 -> 1.450 ms with the last tail moved off it).
            // (Re-placing the tails from per-class end times measured in the first U steps -- each behind the stream that ends
            // first, tried against this plan for a few U steps -- found nothing better on the headline shape and picked worse
            // placements on the Netflix and Yahoo shapes: 61 and 157 ms against 55 and 135 ms per U step.)
            std::vector<int> ring;
            for (int l = 0; l < nlane; ++l) ring.push_back(nlane >= 2 && l < 2 ? l ^ 1 : l);
            std::vector<int> tail_ring;
            for (int l : ring) if (l != 0) { tail_ring.push_back(l); if (tail_ring.size() == 1 && nhead) tail_ring.push_back(MAXLANE); }
            tail_ring.push_back(0);
            for (size_t i = nhead; i < order.size(); ++i) {
                const size_t j = i - nhead;
                plan.push_back({(int)i, j < ring.size() ? ring[j] : tail_ring[(j - ring.size()) % tail_ring.size()]});
            }
        }
        if (tune.debug && !plan_announced && (plan_announced = true))
            for (auto& pr : plan) fprintf(stderr, "[pcr] U step: %s on stream %d\n", pname("ustep", *order[pr.first]).c_str(), pr.second);
        // nothing in flight on the solver's stream (the usual case: the V step has just read its objective back): the lanes
        // need no fork event, their kernels start as soon as they are launched.
        // INVARIANT this rests on: "st idle" means "the solver idle" -- every producer on another stream (the lanes and the
        // high-priority stream here and in for_bins, ar_st in launch_spmm) is joined back INTO st (a wait of st on its event, or a
        // host wait) before the call that launched it returns, so no kernel that writes U, V or the sorted state can still be
        // running on a side stream once st has drained.  Anything that ever launches asynchronously on a lane without joining
        // it into st must record / wait ev_fork here unconditionally.
        const bool idle = hipStreamQuery(st) == hipSuccess;
        ProfScope wall(this, "wall:ustep", st);
        bool used[MAXLANE + 1] = {};
        if (!idle) HIPCHK(hipEventRecord(ev_fork, st));
        for (auto& pr : plan) {
            Bin& b = *order[pr.first];
            hipStream_t q = pr.second == MAXLANE ? hi : lane[pr.second];
            if (!idle && q != st && !used[pr.second]) HIPCHK(hipStreamWaitEvent(q, ev_fork, 0));
            used[pr.second] = true;
            ProfScope ps(this, pname("ustep", b), q, b.nnz, (int64_t)b.users.size());
            launch(b, q);
        }
        for (int l = 1; l < nlane; ++l)
            if (used[l]) { HIPCHK(hipEventRecord(ev_lane[l], lane[l])); RC(join(ev_lane[l])); }
        if (used[MAXLANE]) { HIPCHK(hipEventRecord(ev_hi, hi)); RC(join(ev_hi)); }
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    bool plan_announced = false;
    // the same, back to back on the solver's stream (for kernels shorter than a fork/join round trip)
    template <class F>
    int for_bins_seq(std::vector<Bin>& bs, const char* cls, F launch) {
        for (auto& b : bs) {
            if (b.users.empty()) continue;
            ProfScope ps(this, pname(cls, b), st, b.nnz, (int64_t)b.users.size());
            launch(b, st);
        }
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    // profile slot of one kernel launch: "<class>/<workgroup size>[g]" (g = global-scratch variant)
    // ("ustep" has several classes per workgroup size: "<class>/<workgroup size>.<length bound>")
    static std::string pname(const char* cls, const Bin& b) {
        if (b.gram) return std::string(cls) + "/gram" + std::to_string(b.block) + "." + std::to_string(b.limit);
        std::string s = std::string(cls) + "/" + std::to_string(b.block);
        if (!strcmp(cls, "ustep") && b.limit) s += "." + std::to_string(b.limit);
        // (k_ustep: l = the latency form, 8 rows in flight; r = one-wave class with its rows LDS-resident; #n = symbol id -- together
        // with the workgroup size they name ONE kernel symbol, so a profiler's per-symbol rows can be matched to a class)
        const bool us = !strcmp(cls, "ustep");
        return s + (b.big ? "g" : "") + (b.K > 1 ? "c" : "") + (us && b.K == 1 && b.unr == 8 ? "l" : "") + (us && b.block == 64 && b.rcap > 0 ? "r" : "") +
               (b.sym ? "#" + std::to_string(b.sym) : "");
    }
    std::string ustep_classes() override {
        std::string all;
        for (auto& b : ubins) if (!b.users.empty()) { if (!all.empty()) all += ","; all += pname("ustep", b); }
        return all;
    }
    int class_rows(const std::string& slot, double* v) override {
        for (size_t bi = 0; bi < ubins.size(); ++bi) {
            if (ubins[bi].users.empty() || pname("ustep", ubins[bi]) != slot) continue;
            unsigned long long x = 0;
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(&x, d_rowcnt.p + bi, sizeof x, hipMemcpyDeviceToHost));
            *v = (double)x;
            return PCR_OK;
        }
        pcr_set_error("pcr_solver_counter: no U-step class '" + slot + "'");
        return PCR_ERR_ARG;
    }
    size_t small_common(int block) const { return carve_bytes(geo.ld, sizeof(T)) + carve_bytes(block / PCR_WAVE + 1, 8); }
    int strict() const { return prm.solver_type == PCR_SOLVER_PCR ? 1 : 0; }

    // ------------------------------------------------------------------------------ launches
    // m = V_I u, sort, per-user loss -> objp.  Vm = matrix the scores are taken against.
    // out[z] = U[user(z)] . M[rows[z]] for all local ratings (rating-parallel, balanced)
    int launch_sddmm(const T* M, const int32_t* rows, T* out, const int* skip = nullptr) {
        if (nnz_local == 0) return PCR_OK;
        ProfScope ps(this, "sddmm");
        const T* Umat = d_U.p;
        // tile = consecutive ratings one lane group walks.  64 by default; a shard that needs between one and two rounds of
        // workgroups at 64 gets the smallest tile (a multiple of the 8-row batch) with which ONE round holds it all
        // (ml1m: 96 -- 1224 workgroups on 1280 slots instead of 1836; 1.66 -> 1.64 ms per iteration; 80: 1.68, 128: 1.67)
        if (sddmm_tile == 0) {
            sddmm_tile = 64;
            int per_cu = 0;
            const int ngrp0 = 256 / geo.G;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sddmm<T, 256>, 256, (size_t)ngrp0 * 128 * 8) == hipSuccess && per_cu > 0) {
                const int64_t fit = cdiv(nnz_local, (int64_t)ncu * per_cu * ngrp0);
                if (fit > 64 && fit <= 128) sddmm_tile = (int)((fit + 7) / 8 * 8);
            }
        }
        const int tile = sddmm_tile;
        const int ngrp = 256 / geo.G, span = ngrp * tile;
        const int grid = cdiv(nnz_local, span);
        hipLaunchKernelGGL((k_sddmm<T, 256>), dim3(grid), dim3(256), (size_t)span * 8, st, Umat, M, d_ruser.p, rows, nnz_local, out, geo, tile, skip,
                           (const int32_t*)nullptr, (const int2*)nullptr, (const int32_t*)nullptr);
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    // b[sorted position] = u_user . A[item] for all ratings, walked in the tile-major CSC order of k_spmm (k_sddmm with
    // the roles swapped): for item tables far larger than the L2s.  Chosen by sddmm_by_tiles().
    bool sddmm_by_tiles() const { return sddmm_csc; }
    int launch_sddmm_csc(const T* A, T* out, const int* skip) {
        if (nnz_local == 0) return PCR_OK;
        ProfScope ps(this, "sddmm");
        const int span = (256 / geo.G) * spmm_chunk;
        hipLaunchKernelGGL((k_sddmm<T, 256>), dim3(spmm_blocks), dim3(256), (size_t)span * 8, st, A, d_U.p, d_crow.p, d_cuser.p, nnz_local, out, geo,
                           spmm_chunk, skip, d_c2r.p, d_blk_chunks.p, d_chunk_ptr.p);
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }

    bool prepare_is_single_launch() const {
        // one merged launch on small shards (no fork / join: the classes are shorter than a launch round trip); on large ones the
        // per-class launches side by side win (Netflix shape: 7.9 ms against 8.2 + 2.2 ms for the global-scratch class behind it)
        const bool merged = tune.prepare_merged >= 0 ? tune.prepare_merged != 0 : nnz_local < (int64_t)4000000;
        return merged && !pbins[0].users.empty() && !pbins[1].users.empty();
    }
    int launch_prepare(const T* Vm) {
        RC(launch_sddmm(Vm, d_item.p, d_mcsr.p));
        const Shard<T>& shp = sh;
        auto fn = [&](Bin& b, hipStream_t q) {
            const int nus = (int)b.users.size();
            const int cap_pad = b.big ? host_pow2(b.cap) : b.cap, rsc = b.max_lev + 2;     // (LDS: the sort pads virtually)
            const size_t bigb = prepare_bytes<T>(b.cap, cap_pad, rsc, b.big ? 8 : 4);
            const size_t lds = small_common(b.block) + (b.big ? 0 : bigb);
            const int grid = b.big ? std::min(nus, scratch_blocks) : nus;
#define LP(BL, BG) hipLaunchKernelGGL((k_prepare<T, BL, BG>), dim3(grid), dim3(BL), lds, q, shp, geo, b.d_users.p, nus, d_mcsr.p, b.cap, cap_pad, rsc, d_scratch.p, scratch_stride, strict())
            if (b.big) LP(512, true);
            else if (b.block == 64) LP(64, false);
            else if (b.block == 256) LP(256, false);
            else LP(512, false);
#undef LP
        };
        Bin &ba = pbins[0], &bb = pbins[1];
        if (prepare_is_single_launch()) {
            // both LDS-resident classes (<= 256 ratings: one wave per user; <= 4096: one workgroup) in one launch on the
            // solver's stream (k_prepare_all): no fork / join; only users beyond 4096 ratings take a second launch
            const int na = (int)ba.users.size(), nb = (int)bb.users.size();
            const int cpa = ba.cap, cpb = bb.cap, rsa = ba.max_lev + 2, rsb = bb.max_lev + 2;          // (the sort pads virtually)
            const size_t wb = (small_common(64) + prepare_bytes<T>(ba.cap, cpa, rsa, 4) + 15) & ~(size_t)15;
            // workgroup size of the launch = the long users' teams: 512 threads (1024 measured slower: NOTES.md)
            const int wbs = 512, wpb = wbs / 64;
            const size_t lds = std::max(wb * wpb, small_common(wbs) + prepare_bytes<T>(bb.cap, cpb, rsb, 4));
            {
                ProfScope ps(this, "prepare/all", st, ba.nnz + bb.nnz, (int64_t)(na + nb));
                hipLaunchKernelGGL((k_prepare_all<T, 512>), dim3(nb + cdiv(na, wpb)), dim3(512), lds, st, shp, ba.d_users.p, na, ba.cap, cpa, rsa, wb,
                                   bb.d_users.p, nb, bb.cap, cpb, rsb, nb, d_mcsr.p, strict());
            }
            if (!pbins[2].users.empty()) { ProfScope ps(this, pname("prepare", pbins[2]), st, pbins[2].nnz, (int64_t)pbins[2].users.size()); fn(pbins[2], st); }
            HIPCHK(hipGetLastError());
            have_sorted = true;
            sh.prev_valid = 1;
            return PCR_OK;
        }
        RC(for_bins(bins, "prepare", fn));
        have_sorted = true;
        sh.prev_valid = 1;
        return PCR_OK;
    }

    int launch_vsweep(bool hv, const T* A, const int* skip = nullptr) {
        // b = u_user . A_item per rating: walking the sorted state's item ids leaves it in sorted order (what the sweep reads);
        // the CSC walk (item tables beyond the L2s) leaves it in CSR order and the sweep picks it up through sidx
        if (hv) { if (sddmm_by_tiles()) RC(launch_sddmm_csc(A, d_b.p, skip)); else RC(launch_sddmm(A, d_sitem.p, d_b.p, skip)); }
        if (hv && vblock_nbp) {                                   // the block's share of b = U A^T on the matrix cores (pcr_vblock.h)
            ProfScope ps(this, "vblock_b");
            const int64_t tiles = (int64_t)(vblock_nbp / GramMfma<T>::TS) * cdiv(d2, GramMfma<T>::TS);
#ifndef PCR_NO_OPTIONAL_KERNELS
            hipLaunchKernelGGL((k_vblock_b<T>), dim3((unsigned)cdiv(tiles, 4)), dim3(256), 0, st, d_U.p, A, d_blk_user.p, vblock_nbp, d_cpos_dense.p, d2, geo, d_b.p, skip);
#endif
            HIPCHK(hipGetLastError());
        }
        return launch_sweeps(hv, skip, hv && sddmm_by_tiles());
    }
    // the per-user sweeps alone: b (d_b) -> c (CSR order, d_c)
    int launch_sweeps(bool hv, const int* skip, bool b_csr = false) {
        const int bc = (b_csr ? 1 : 0) | (sweep_pf4 ? 2 : 0);
        const bool two = hv && !sh.ws;                      // scores and sweep values both live in LDS (no window cache)
        auto fn = [&](Bin& b, hipStream_t q) {
            const int nus = (int)b.users.size();
            const int rsc = b.max_lev + 2;
            if (b.block == 64) {                         // short users: one wave each, four per workgroup
                const size_t wb = (vsweep_wave_bytes<T>(b.cap, rsc, two) + 15) & ~(size_t)15;
                if (hv) hipLaunchKernelGGL((k_vsweep_wave<T, true>), dim3(cdiv(nus, 4)), dim3(256), wb * 4, q, sh, b.d_users.p, nus, d_b.p, d_c.p, b.cap, rsc, wb, strict(), skip, bc);
                else hipLaunchKernelGGL((k_vsweep_wave<T, false>), dim3(cdiv(nus, 4)), dim3(256), wb * 4, q, sh, b.d_users.p, nus, d_b.p, d_c.p, b.cap, rsc, wb, strict(), skip, bc);
                return;
            }
            const size_t bigb = vsweep_bytes<T>(b.cap, rsc, two);
            const size_t lds = small_common(b.block) + (b.big ? 0 : bigb);
            const int grid = b.big ? std::min(nus, scratch_blocks) : nus;
#define LV(BL, BG, HV) hipLaunchKernelGGL((k_vsweep<T, BL, BG, HV>), dim3(grid), dim3(BL), lds, q, sh, geo, b.d_users.p, nus, d_b.p, d_c.p, b.cap, rsc, d_scratch.p, scratch_stride, strict(), skip, bc)
            if (hv) { if (b.big) LV(512, true, true); else LV(512, false, true); }
            else { if (b.big) LV(512, true, false); else LV(512, false, false); }
#undef LV
        };
        Bin &ba = sbins[0], &bb = sbins[1];
        if (!ba.users.empty() && !bb.users.empty()) {
            // the two LDS-resident classes in one launch (k_vsweep_all); only users beyond 4096 ratings take a second one
            const int na = (int)ba.users.size(), nb = (int)bb.users.size();
            const int rsa = ba.max_lev + 2, rsb = bb.max_lev + 2;
            const size_t wb = (vsweep_wave_bytes<T>(ba.cap, rsa, two) + 15) & ~(size_t)15;
            const int wbs = 512, wpb = wbs / 64;
            const size_t lds = std::max(wb * wpb, small_common(wbs) + vsweep_bytes<T>(bb.cap, rsb, two));
            const int grid = nb + cdiv(na, wpb);
            {
                ProfScope ps(this, std::string(hv ? "vhv" : "vgrad") + "/all", st, ba.nnz + bb.nnz, (int64_t)(na + nb));
#define LVA(HV, WBS, MW) hipLaunchKernelGGL((k_vsweep_all<T, HV, WBS, MW>), dim3(grid), dim3(WBS), lds, st, sh, ba.d_users.p, na, ba.cap, rsa, wb, \
                                        bb.d_users.p, nb, bb.cap, rsb, nb, d_b.p, d_c.p, strict(), skip, bc)
                // (four workgroups per CU where the longest user's arrays fit a quarter of the LDS: the 64-VGPR symbol)
                const bool dense = lds <= (size_t)40 * 1024;
                if (hv) { if (dense) LVA(true, 512, 8); else LVA(true, 512, 1); } else { if (dense) LVA(false, 512, 8); else LVA(false, 512, 1); }
#undef LVA
            }
            if (!sbins[2].users.empty()) { ProfScope ps(this, pname(hv ? "vhv" : "vgrad", sbins[2]), st, sbins[2].nnz, (int64_t)sbins[2].users.size()); fn(sbins[2], st); }
            HIPCHK(hipGetLastError());
            return PCR_OK;
        }
        RC(for_bins_seq(sbins, hv ? "vhv" : "vgrad", fn));
        return PCR_OK;
    }

    // out = beta * base + sum c * U-rows (item-major, deterministic slab reduction)
    // dots_rr != nullptr: also leave the partials of base.out and dots_rr.base in d_partA (k_spmm_fin DOTS)
    int fin_blocks() const { return (int)std::min<int64_t>(1024, cdiv(d2, 256 / geo.G)); }
    // one item range of the SpMM: k_spmm over the range's workgroups, k_spmm_fin over its items
    int launch_spmm_range(int r, T* out, const T* base, double beta, const int* skip, const T* dots_rr) {
        if (nnz_local > 0 && rng_blk[r + 1] > rng_blk[r]) {
            ProfScope ps(this, "spmm");
            hipLaunchKernelGGL((k_spmm<T, 256>), dim3(rng_blk[r + 1] - rng_blk[r]), dim3(256), 0, st, d_c.p, d_c2r.p, d_cuf.p,
                               d_chunk_ptr.p, d_slot_base.p, d_slot_id.p, d_blk_chunks.p + rng_blk[r], d_U.p, d_slab.p, geo, skip);
        }
        const int j0 = (int)rng_item[r], j1 = (int)rng_item[r + 1];
        if (j1 <= j0) return PCR_OK;
        const int grid = n_rng == 1 ? fin_blocks() : (int)std::min<int64_t>(1024, cdiv(j1 - j0, 256 / geo.G));
        ProfScope ps2(this, "spmm_fin");
        if (dots_rr) hipLaunchKernelGGL((k_spmm_fin<T, 256, true>), dim3(grid), dim3(256), 0, st, d_slab.p, d_item_slot.p, base, beta, j1, out, geo, skip, dots_rr, d_partA.p, j0);
        else hipLaunchKernelGGL((k_spmm_fin<T, 256, false>), dim3(grid), dim3(256), 0, st, d_slab.p, d_item_slot.p, base, beta, j1, out, geo, skip, (const T*)nullptr, (double*)nullptr, j0);
        return PCR_OK;
    }
    // out = beta * base + sum c * U-rows (item-major, deterministic slab reduction), summed over the ranks.
    // dots_rr != nullptr (one GPU, one range): also leave the partials of base.out and dots_rr.base in d_partA (k_spmm_fin DOTS)
    // With item ranges (n_rng > 1) the all-reduce of range r runs on its own stream while the SpMM of range r + 1 computes: the
    // ranges are queued one ahead of the exchange, so this holds for the host-driven peer-to-peer exchange as for RCCL.
    int launch_spmm(T* out, const T* base, double beta, const int* skip = nullptr, const T* dots_rr = nullptr) {
        const size_t n = (size_t)d2 * geo.ld;
        if (n_rng == 1) {
            RC(launch_spmm_range(0, out, base, beta, skip, dots_rr));
            if (vblock_nbp) {                                     // + the block's share, C_B^T U_B on the matrix cores, before the exchange
                ProfScope ps(this, "vblock_hp");
                const int64_t tiles = (int64_t)cdiv(d2, GramMfma<T>::TS) * cdiv(geo.ld, GramMfma<T>::TS);
#ifndef PCR_NO_OPTIONAL_KERNELS
                hipLaunchKernelGGL((k_vblock_hp<T>), dim3((unsigned)cdiv(tiles, 4)), dim3(256), 0, st, d_U.p, d_c.p, d_blk_user.p, vblock_nbp, d_cpos_dense.p, d2, geo, out, skip);
#endif
            }
            HIPCHK(hipGetLastError());
            return allreduce_T(out, n);
        }
        const bool exch = !single();
        auto queue = [&](int r) -> int {
            RC(launch_spmm_range(r, out, base, beta, skip, nullptr));
            if (exch) HIPCHK(hipEventRecord(ev_rng[r], st));
            return PCR_OK;
        };
        RC(queue(0));
        for (int r = 0; r < n_rng; ++r) {
            if (r + 1 < n_rng) RC(queue(r + 1));
            if (!exch) continue;
            HIPCHK(hipStreamWaitEvent(ar_st, ev_rng[r], 0));
            const size_t lo = (size_t)rng_item[r] * geo.ld, hi = (size_t)rng_item[r + 1] * geo.ld;
            if (hi > lo) RC(allreduce_T(out + lo, hi - lo, ar_st));
        }
        if (exch) { HIPCHK(hipEventRecord(ev_ar, ar_st)); HIPCHK(hipStreamWaitEvent(st, ev_ar, 0)); }
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }

    int allreduce_T(T* buf, size_t count, hipStream_t q = nullptr) {
        if (single()) return PCR_OK;
        if (!comm && !p2p) { pcr_set_error("nranks > 1 but neither pcr_solver_comm_init nor pcr_solver_comm_init_p2p was called"); return PCR_ERR_STATE; }
        if (!q) q = st;
        ProfScope ps(this, "allreduce", q);
        if (p2p) {
            if (!p2p->allreduce<T>(buf, count, q)) { pcr_set_error("p2p all-reduce: " + p2p->err); return PCR_ERR_COMM; }
            return PCR_OK;
        }
        NCCLCHK(ncclAllReduce(buf, buf, count, sizeof(T) == 4 ? ncclFloat : ncclDouble, ncclSum, comm, q));
        return PCR_OK;
    }
    int allreduce_f64(double* buf, size_t count) {
        if (single()) return PCR_OK;
        if (!comm && !p2p) { pcr_set_error("nranks > 1 but neither pcr_solver_comm_init nor pcr_solver_comm_init_p2p was called"); return PCR_ERR_STATE; }
        ProfScope ps(this, "allreduce");
        if (p2p) {
            if (!p2p->allreduce<double>(buf, count, st, true)) { pcr_set_error("p2p all-reduce: " + p2p->err); return PCR_ERR_COMM; }
            return PCR_OK;
        }
        NCCLCHK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, st));
        return PCR_OK;
    }

    // deterministic sum of a double array into d_scal[slot] (two-stage)
    int reduce_sum(const double* in, int64_t n, int slot) {
        const int nb = (int)std::min<int64_t>(512, std::max<int64_t>(1, cdiv(n, 2048)));
        const int per = cdiv(std::max<int64_t>(n, 1), nb);
        hipLaunchKernelGGL(k_sum_stage1, dim3(nb), dim3(PCR_EW_BLOCK), 0, st, in, n, per, d_partA.p);
        hipLaunchKernelGGL(k_fin2, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partA.p, nb, d_scal.p + slot);
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    // |a|^2 of a T array into d_scal[slot] (slot+1 receives dot(a,b) or 0)
    int norm2(const T* a, int64_t n, int slot) {
        const int nb = (int)std::min<int64_t>(512, std::max<int64_t>(1, cdiv(n, 4096)));
        const int per = cdiv(std::max<int64_t>(n, 1), nb);
        hipLaunchKernelGGL((k_dots<T>), dim3(nb), dim3(PCR_EW_BLOCK), 0, st, a, (const T*)nullptr, n, per, d_partB.p);
        hipLaunchKernelGGL(k_fin2, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partB.p, nb, d_scal.p + slot);
        HIPCHK(hipGetLastError());
        return PCR_OK;
    }
    // Every host read of all-reduced data goes through this: wait for the stream, then ask whether a device-driven exchange
    // ran into its deadline on the way (the data would be garbage; the job's shared error flag is raised for the peers).
    int sync_checked() {
        HIPCHK(hipStreamSynchronize(st));
        if (p2p && p2p->exchange_failed()) { pcr_set_error("p2p all-reduce: " + p2p->err); return PCR_ERR_COMM; }
        return PCR_OK;
    }
    int fetch_scal(int count) {
        HIPCHK(hipMemcpyAsync(h_scal, d_scal.p, std::max(count, 44) * sizeof(double), hipMemcpyDeviceToHost, st));   // [32..44): the CG scalars
        return sync_checked();
    }

    // loss (all ranks) of the last prepare + lambda/2 (|U|^2 + |Vm|^2)   (pcrpp.cpp:410)
    // d_scal[0] = sum objx (all ranks), [1] = |Vm|^2, [2] = |U|^2 (all ranks; only if with_u): one pass + one finish
    // objx2 (optional): a second per-user sum -> [3].  after_ustep: the finishing kernel also moves the U step's counters to
    // d_scal[slot + 4 .. slot + 7) and resets the counter block (k_fin4).
    int objective_sums(const double* objx, const T* Vm, bool with_u, int slot = 0, const double* objx2 = nullptr, bool after_ustep = false) {
        const int64_t nV = (int64_t)d2 * geo.ld, nU = (int64_t)n_users * geo.ld;
        const int nb = (int)std::min<int64_t>(512, std::max<int64_t>(1, cdiv(std::max(nV, nU), 4096)));
        hipLaunchKernelGGL((k_obj3<T>), dim3(nb), dim3(PCR_EW_BLOCK), 0, st, objx, objx2, n_users, Vm, nV, with_u ? d_U.p : (const T*)nullptr, nU, d_partA.p);
        // N ranks: ONE all-reduce of the four columns -- the per-user sums and |U|^2 are shard partials, |Vm|^2 (replicated) is
        // contributed by rank 0 alone
        const bool reduce = !single();
        const int keep1 = (!reduce || rank == 0) ? 1 : 0;
        if (after_ustep) hipLaunchKernelGGL(k_fin4, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partA.p, nb, d_scal.p + slot, d_counters.p, d_scal.p + slot + 4, counter_words(), keep1);
        else hipLaunchKernelGGL(k_fin4, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partA.p, nb, d_scal.p + slot, (unsigned long long*)nullptr, (double*)nullptr, 0, keep1);
        HIPCHK(hipGetLastError());
        // (after a U step the shard's CG / line-search counts and its cluster time-out flag ride on the same all-reduce:
        // every rank then sees the job's totals and the same error, and all of them leave the loop together)
        RC(allreduce_f64(d_scal.p + slot, after_ustep ? 8 : 4));
        return PCR_OK;
    }
    int full_objective(const T* Vm, double* obj) {
        const bool need_u = !unorm_valid;                          // |U|^2 rides on the same pass and read-back
        RC(objective_sums(d_objp.p, Vm, need_u));
        RC(fetch_scal(8));                                         // slots 4..6: update_V's starting objective, queued earlier
        if (need_u) { unorm2 = h_scal[2]; unorm_valid = true; }
        *obj = h_scal[0] + prm.lambda * (unorm2 + h_scal[1]) / 2.0;
        return PCR_OK;
    }

    // ------------------------------------------------------------------------------ host <-> device
    // Factor matrices cross the boundary as the reference's fp64 row-major payload (mat_t); the device keeps rows padded to ld
    // elements of T.  The conversion runs on the device (k_mat_in / k_mat_out) on slabs of at most 64 M values, straight from / into
    // the caller's buffer: no host-side staging copy, no serial conversion loop (48 M values at the Netflix shape).
    int upload_mat(const double* H, int64_t rows, T* D) {
        const int64_t slab_rows = std::max<int64_t>(1, ((int64_t)64 << 20) / std::max(1, geo.r));
        DBuf<double> stage;
        RC(stage.alloc((size_t)std::min(rows, slab_rows) * geo.r));
        for (int64_t r0 = 0; r0 < rows; r0 += slab_rows) {
            const int64_t nr = std::min(slab_rows, rows - r0);
            HIPCHK(hipMemcpyAsync(stage.p, H + r0 * geo.r, (size_t)nr * geo.r * sizeof(double), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((k_mat_in<T>), dim3((unsigned)std::min<int64_t>(1 << 16, cdiv(nr * geo.ld, 256))), dim3(256), 0, st, stage.p, D + r0 * geo.ld, nr, geo.r, geo.ld);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
        return PCR_OK;
    }
    int download_mat(const T* D, int64_t rows, double* H) {
        const int64_t slab_rows = std::max<int64_t>(1, ((int64_t)64 << 20) / std::max(1, geo.r));
        DBuf<double> stage;
        RC(stage.alloc((size_t)std::min(rows, slab_rows) * geo.r));
        for (int64_t r0 = 0; r0 < rows; r0 += slab_rows) {
            const int64_t nr = std::min(slab_rows, rows - r0);
            hipLaunchKernelGGL((k_mat_out<T>), dim3((unsigned)std::min<int64_t>(1 << 16, cdiv(nr * geo.r, 256))), dim3(256), 0, st, D + r0 * geo.ld, stage.p, nr, geo.r, geo.ld);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(H + r0 * geo.r, stage.p, (size_t)nr * geo.r * sizeof(double), hipMemcpyDeviceToHost, st));
            RC(sync_checked());
        }
        if (rows == 0) RC(sync_checked());
        return PCR_OK;
    }

    // local: U holds this rank's n_users rows only (else the full d1 x k matrix, of which this rank touches its own rows)
    int set_factors(const double* U, const double* V, bool local) override {
        if (U) { RC(upload_mat(U + (local ? 0 : first_user * geo.r), n_users, d_U.p)); unorm_valid = false; }
        if (V) RC(upload_mat(V, d2, d_V.p));
        have_sorted = false; state_of_rejected_V = false;
        return PCR_OK;
    }
    int get_factors(double* U, double* V, bool local) override {
        if (U) RC(download_mat(d_U.p, n_users, U + (local ? 0 : first_user * geo.r)));
        if (V) RC(download_mat(d_V.p, d2, V));
        return PCR_OK;
    }

    // ------------------------------------------------------------------------------ per-function ABI
    int comp_m(double* m_out) override {
        T* mc = d_mcsr.p;
        RC(launch_prepare(d_V.p));
        if (m_out) {
            std::vector<T> tmp(nnz_local);
            if (nnz_local) HIPCHK(hipMemcpyAsync(tmp.data(), mc, nnz_local * sizeof(T), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int64_t z = 0; z < nnz_local; ++z) m_out[z] = (double)tmp[z];
        }
        return PCR_OK;
    }
    int need_sorted() const {
        if (!have_sorted) { pcr_set_error("call pcr_comp_m first (no sorted state for the current factors)"); return PCR_ERR_STATE; }
        return PCR_OK;
    }
    int objective(double* obj) override {
        RC(need_sorted());
        return full_objective(d_V.p, obj);
    }

    // g = lambda V + sum_i sum_j c_ij u_i   (pcrpp.cpp:140-249) into d_g
    int device_gradient() {
        RC(launch_vsweep(false, nullptr));
        return launch_spmm(d_g.p, d_V.p, rank == 0 ? prm.lambda : 0.0);    // rank 0 carries the lambda*V term; summed over the ranks
    }
    int obtain_g(double* g) override {
        RC(need_sorted());
        RC(device_gradient());
        return download_mat(d_g.p, d2, g);
    }
    // out = lambda p + sum c(b) u   (the lambda term on rank 0 only; summed by the all-reduce)
    int device_hv(const T* pvec, T* out, const int* skip = nullptr, const T* dots_rr = nullptr) {
        RC(launch_vsweep(true, pvec, skip));
        return launch_spmm(out, pvec, rank == 0 ? prm.lambda : 0.0, skip, dots_rr);
    }
#ifdef PCR_PIPE_PROBE
    // Developer build only (make lib LIBDIR=build_next/probe PCR_EXTRA=-DPCR_PIPE_PROBE; tools/exp_pipe_probe.py): TIMING of one
    // Hessian-vector product as it runs today (five kernels back to back on the solver's stream) against the two-stream form of
    // VERDICT r5 item 7 -- each of the three tile-parallel kernels over HALF the ratings / users / workgroups on either stream,
    // one device-side fork and one join per product, k_spmm_fin behind the join.  The halves are cut by position (the first and
    // second half of the rating range, of both user lists, of the SpMM's workgroups), which is what a tile-wise split would launch;
    // the second halves read what the first halves have not finished writing, so the VALUES are meaningless -- only the clock counts.
    int pipe_probe() {
        if (sbins[0].users.empty() || sbins[1].users.empty() || n_rng != 1 || nnz_local < 4096) return PCR_OK;
        hipStream_t sd = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
        hipEvent_t e0, e1, ef, ej;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        const int tile = sddmm_tile ? sddmm_tile : 64, ngrp = 256 / geo.G, span = ngrp * tile;
        Bin &ba = sbins[0], &bb = sbins[1];
        const int na = (int)ba.users.size(), nb = (int)bb.users.size(), rsa = ba.max_lev + 2, rsb = bb.max_lev + 2;
        const bool two = !sh.ws;
        const size_t wb = (vsweep_wave_bytes<T>(ba.cap, rsa, two) + 15) & ~(size_t)15;
        const size_t lds = std::max(wb * 8, small_common(512) + vsweep_bytes<T>(bb.cap, rsb, two));
        const bool dense = lds <= (size_t)40 * 1024;
        const int bc = sweep_pf4 ? 2 : 0;
        auto sddmm = [&](hipStream_t q, int64_t z0, int64_t z1) {
            hipLaunchKernelGGL((k_sddmm<T, 256>), dim3((unsigned)cdiv(z1 - z0, span)), dim3(256), (size_t)span * 8, q, d_U.p, d_p.p, d_ruser.p + z0, d_sitem.p + z0, z1 - z0,
                               d_b.p + z0, geo, tile, (const int*)nullptr, (const int32_t*)nullptr, (const int2*)nullptr, (const int32_t*)nullptr);
        };
        auto sweep = [&](hipStream_t q, int a0, int a1, int b0, int b1) {
            const int grid = (b1 - b0) + cdiv(a1 - a0, 8);
            if (dense) hipLaunchKernelGGL((k_vsweep_all<T, true, 512, 8>), dim3(grid), dim3(512), lds, q, sh, ba.d_users.p + a0, a1 - a0, ba.cap, rsa, wb, bb.d_users.p + b0, b1 - b0, bb.cap, rsb, b1 - b0, d_b.p, d_c.p, strict(), (const int*)nullptr, bc);
            else hipLaunchKernelGGL((k_vsweep_all<T, true, 512, 1>), dim3(grid), dim3(512), lds, q, sh, ba.d_users.p + a0, a1 - a0, ba.cap, rsa, wb, bb.d_users.p + b0, b1 - b0, bb.cap, rsb, b1 - b0, d_b.p, d_c.p, strict(), (const int*)nullptr, bc);
        };
        auto spmm = [&](hipStream_t q, int w0, int w1) {
            hipLaunchKernelGGL((k_spmm<T, 256>), dim3(w1 - w0), dim3(256), 0, q, d_c.p, d_c2r.p, d_cuf.p, d_chunk_ptr.p, d_slot_base.p, d_slot_id.p, d_blk_chunks.p + w0, d_U.p, d_slab.p, geo, (const int*)nullptr);
        };
        auto fin = [&](hipStream_t q) {
            hipLaunchKernelGGL((k_spmm_fin<T, 256, false>), dim3(fin_blocks()), dim3(256), 0, q, d_slab.p, d_item_slot.p, d_p.p, 0.0, (int)d2, d_Hp.p, geo, (const int*)nullptr, (const T*)nullptr, (double*)nullptr, 0);
        };
        const int64_t zh = nnz_local / 2;
        const int wh = (spmm_blocks / 2) & ~7;
        const int R = 60;
        auto timed = [&](const char* what, auto body) -> int {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipStreamSynchronize(sd));
                HIPCHK(hipEventRecord(e0, st));
                for (int i = 0; i < R; ++i) body();
                HIPCHK(hipEventRecord(e1, st));
                HIPCHK(hipEventSynchronize(e1));
                HIPCHK(hipStreamSynchronize(sd));
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            fprintf(stderr, "[pipe-probe] %-74s %7.2f us per product\n", what, 1e3 * best / R);
            return PCR_OK;
        };
        RC(timed("today: sddmm, sweep, spmm, fin on one stream", [&]() { sddmm(st, 0, nnz_local); sweep(st, 0, na, 0, nb); spmm(st, 0, spmm_blocks); fin(st); }));
        RC(timed("  sddmm alone", [&]() { sddmm(st, 0, nnz_local); }));
        RC(timed("  sddmm, first half of the ratings alone", [&]() { sddmm(st, 0, zh); }));
        RC(timed("  sweep alone", [&]() { sweep(st, 0, na, 0, nb); }));
        RC(timed("  sweep, first half of both user lists (the longest users) alone", [&]() { sweep(st, 0, na / 2, 0, nb / 2); }));
        RC(timed("  spmm alone", [&]() { spmm(st, 0, spmm_blocks); }));
        RC(timed("  spmm, first half of the workgroups alone", [&]() { spmm(st, 0, wh); }));
        RC(timed("  fin alone", [&]() { fin(st); }));
        RC(timed("halves back to back on ONE stream (what the split itself costs)", [&]() {
            sddmm(st, 0, zh); sddmm(st, zh, nnz_local); sweep(st, 0, na / 2, 0, nb / 2); sweep(st, na / 2, na, nb / 2, nb); spmm(st, 0, wh); spmm(st, wh, spmm_blocks); fin(st); }));
        RC(timed("two streams: halves side by side, fork + join per product, fin behind the join", [&]() {
            (void)hipEventRecord(ef, st); (void)hipStreamWaitEvent(sd, ef, 0);
            sddmm(st, 0, zh); sddmm(sd, zh, nnz_local); sweep(st, 0, na / 2, 0, nb / 2); sweep(sd, na / 2, na, nb / 2, nb); spmm(st, 0, wh); spmm(sd, wh, spmm_blocks);
            (void)hipEventRecord(ej, sd); (void)hipStreamWaitEvent(st, ej, 0);
            fin(st); }));
        RC(timed("two streams, the side stream one kernel behind (its sddmm starts with the other half's sweep)", [&]() {
            sddmm(st, 0, zh);
            (void)hipEventRecord(ef, st); (void)hipStreamWaitEvent(sd, ef, 0);
            sweep(st, 0, na / 2, 0, nb / 2); sddmm(sd, zh, nnz_local); spmm(st, 0, wh); sweep(sd, na / 2, na, nb / 2, nb); spmm(sd, wh, spmm_blocks);
            (void)hipEventRecord(ej, sd); (void)hipStreamWaitEvent(st, ej, 0);
            fin(st); }));
        RC(timed("full-size sweep on the side stream beside the full-size sddmm (latency-bound beside bandwidth-bound)", [&]() {
            (void)hipEventRecord(ef, st); (void)hipStreamWaitEvent(sd, ef, 0);
            sddmm(st, 0, nnz_local); sweep(sd, 0, na, 0, nb);
            (void)hipEventRecord(ej, sd); (void)hipStreamWaitEvent(st, ej, 0); }));
        RC(timed("full-size spmm on the side stream beside the full-size sddmm (two bandwidth-bound kernels)", [&]() {
            (void)hipEventRecord(ef, st); (void)hipStreamWaitEvent(sd, ef, 0);
            sddmm(st, 0, nnz_local); spmm(sd, 0, spmm_blocks);
            (void)hipEventRecord(ej, sd); (void)hipStreamWaitEvent(st, ej, 0); }));
        HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipStreamSynchronize(sd));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ef); (void)hipEventDestroy(ej); (void)hipStreamDestroy(sd);
        return PCR_OK;
    }
#endif
    int compute_Ha(const double* a, double* Ha) override {
        RC(need_sorted());
        RC(upload_mat(a, d2, d_p.p));
#ifdef PCR_PIPE_PROBE
        RC(pipe_probe());
        RC(need_sorted());
#endif
        RC(device_hv(d_p.p, d_Hp.p));
        return download_mat(d_Hp.p, d2, Ha);
    }

    // CG on H delta = g with g in d_g (pcrpp.cpp:335-358); result in d_delta
    // iters == nullptr: no host sync; the iteration count is in h_cg once the stream has passed this point
    int device_cg(int* iters) {
        const int64_t n = (int64_t)d2 * geo.ld;
        {
            ProfScope ps(this, "cg");
            hipLaunchKernelGGL((k_cg_init<T>), dim3(ew_blocks), dim3(PCR_EW_BLOCK), 0, st, d_g.p, d_delta.p, d_rr.p, d_p.p, n, ew_per_block, d_partA.p);
            hipLaunchKernelGGL(k_cg_init_fin, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partA.p, ew_blocks, d_cgp, prm.cg_tol);
        }
        // All 10 iterations are queued without a host round trip; once the device-side stop test
        // (pcrpp.cpp:350) fires, the remaining kernels return immediately.  One sync at the end.
        const int* skip = &d_cgp->done;
        // one GPU: Hp is final when k_spmm_fin stores it, so that kernel also produces the p.Hp / rr.p partials;
        // with an all-reduce in between they need their own pass (k_cg_a)
        const bool fused_dots = single() && n_rng == 1 && !vblock_nbp;      // (the dense block adds to Hp AFTER k_spmm_fin)
        const bool exact_rr = prm.cg_tol < 1e-5;                   // the residual recurrence of k_cg_bc cancels below that
        for (int k = 1; k <= prm.cg_max_iter; ++k) {
            RC(device_hv(d_p.p, d_Hp.p, skip, fused_dots ? d_rr.p : nullptr));
            {
                ProfScope ps(this, "cg");
                if (!fused_dots) hipLaunchKernelGGL((k_cg_a<T>), dim3(ew_blocks), dim3(PCR_EW_BLOCK), 0, st, d_p.p, d_Hp.p, d_rr.p, n, ew_per_block, d_partA.p, d_cgp);
                if (exact_rr) {
                    hipLaunchKernelGGL((k_cg_bc<T, true>), dim3(ew_blocks), dim3(PCR_EW_BLOCK), 0, st, d_p.p, d_Hp.p, d_rr.p, d_delta.p, n, ew_per_block, fused_dots ? fin_blocks() : ew_blocks, d_partA.p, d_cgp, k, d_partB.p);
                    hipLaunchKernelGGL(k_cg_stop, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partB.p, ew_blocks, d_cgp, k);
                } else {
                    hipLaunchKernelGGL((k_cg_bc<T, false>), dim3(ew_blocks), dim3(PCR_EW_BLOCK), 0, st, d_p.p, d_Hp.p, d_rr.p, d_delta.p, n, ew_per_block, fused_dots ? fin_blocks() : ew_blocks, d_partA.p, d_cgp, k, (double*)nullptr);
                }
            }
            HIPCHK(hipGetLastError());
        }
        if (iters) {
            HIPCHK(hipMemcpyAsync(h_cg, d_cgp, sizeof(CGState), hipMemcpyDeviceToHost, st));
            RC(sync_checked());
            *iters = h_cg->iters;
        }
        return PCR_OK;
    }
    int solve_delta(const double* g, double* delta, int* iters) override {
        RC(need_sorted());
        RC(upload_mat(g, d2, d_g.p));
        RC(device_cg(iters));
        return download_mat(d_delta.p, d2, delta);
    }

    // pcrpp.cpp:415-444
    int update_V(double* now_obj, int* info) override {
        int cg_iters = 0, tries = 0, accepted = 0;
        if (!have_sorted) RC(launch_prepare(d_V.p));      // comp_m_new (:417); after a U step its line search left exactly this state
        double prev_obj = 0.0;
        // objective_new(m, U, V) (:425) of the starting point: its three sums are QUEUED here -- by the same kernels, in the
        // same summation order as the objectives of the line search, so that "no strict decrease" (q5) compares like with
        // like exactly as in the reference -- and read back together with the first line-search objective: no host round
        // trip of its own.  (After a U step the per-user losses in objp are the ones k_ustep left.)
        const bool prev_u = !unorm_valid;
        const double unorm2_before = unorm2;
        if (start_sums_queued) start_sums_queued = false;          // ... already, by the U step this V step follows (prev_u is true then)
        else RC(objective_sums(d_objp.p, d_V.p, prev_u, 4));
        RC(device_gradient());                                     // obtain_g_new (:418)
        // solve_delta_new (:422).  No host round trip when the line search's prepare is the single-launch form: everything
        // up to the objective read-back is stream-ordered.  (With the per-class launches the host waits for the CG first: a
        // fork whose event is still pending when the lanes reach it costs far more here than the round trip -- measured
        // 2.19 -> 2.40 ms per iteration.)
        const bool cg_sync = !prepare_is_single_launch();
        RC(device_cg(cg_sync ? &cg_iters : nullptr));
        double step = prm.stepsize, obj = prev_obj;
        const int64_t n = (int64_t)d2 * geo.ld;
        for (int it = 0; it < 20; ++it) {                          // :427-441
            hipLaunchKernelGGL((k_axpy_out<T>), dim3(cdiv(n, 256)), dim3(256), 0, st, d_Vnew.p, d_V.p, d_delta.p, -step, n);
            RC(launch_prepare(d_Vnew.p));
            RC(full_objective(d_Vnew.p, &obj));
            if (!cg_sync) cg_iters = h_cg->iters;                 // that read-back synchronised the stream
            if (it == 0) prev_obj = h_scal[4] + prm.lambda * ((prev_u ? h_scal[6] : unorm2_before) + h_scal[5]) / 2.0;
            if (ustep_pending) {                                   // ... and with it the U step queued before this V step
                RC(ustep_finish(&fin_obj, fin_info));
                fin_ready = true;
            }
            ++tries;
            if (obj < prev_obj) {
                std::swap(d_V.p, d_Vnew.p);
                accepted = 1;
                break;
            }
            step /= 2.0;
        }
        // the sorted state now belongs to the LAST TRIED V_new, accepted or not (:430-431, :443)
        state_of_rejected_V = !accepted;
        if (now_obj) *now_obj = obj;
        if (info) { info[0] = cg_iters; info[1] = tries; info[2] = accepted; }
        return PCR_OK;
    }

    // counters + cluster barriers of the next U step: reset by the kernel that finishes the previous U step's sums (k_fin4),
    // by a memset only before the first one -- the U step then finds the solver's stream idle and starts without a fork
    // (for_ubins)
    bool counters_zeroed = false;
    int counter_words() const { return 4 + 64 + (bar_n + 1) / 2; }
    int zero_counters() {
        if (counters_zeroed) return PCR_OK;
        HIPCHK(hipMemsetAsync(d_counters.p, 0, (size_t)counter_words() * sizeof(unsigned long long), st));
        counters_zeroed = true;
        return PCR_OK;
    }
    int launch_ustep() {
        RC(zero_counters());
        counters_zeroed = false;
        // pcr_tune("ustep_newton"): exact Newton directions first (explicit Hessian + Cholesky, k_unewton); every other user's CG
        // runs to convergence instead of stopping after cg_max_iter iterations -- the same step by another route
        const bool newton = tune.ustep_newton != 0;
        const double* dirp = nullptr;
        if (newton && d_dir.p) {
            HIPCHK(hipMemsetAsync(d_dir.p, 0xFF, d_dir.n * sizeof(double), st));          // all ones = NaN: "no direction"
            if (newton_n > 0) {
                const int ldp = (geo.ld + 15) & ~15;
                ProfScope ps(this, "unewton", st, -1, newton_n);
                // (the window bounds ride along where the workgroup still fits half a CU's LDS: two workgroups per CU matter more)
                const int wbcap = newton_bytes<T>(newton_cap, newton_rs, geo.ld, ldp, newton_cap) <= (size_t)80 * 1024 ? newton_cap : 0;
                const size_t lds = newton_bytes<T>(newton_cap, newton_rs, geo.ld, ldp, wbcap);
#ifndef PCR_NO_OPTIONAL_KERNELS
                hipLaunchKernelGGL((k_unewton<T>), dim3(newton_grid), dim3(256), lds, st, sh, geo,
                                   d_newton_users.p, newton_n, d_U.p, d_V.p, prm.lambda, strict(), newton_cap, newton_rs, ldp, d_newton_scratch.p, newton_stride, d_dir.p, wbcap);
#else
                (void)lds;
#endif
                HIPCHK(hipGetLastError());
            }
            dirp = d_dir.p;
        }
        const int cg_max_u = newton ? 2 * geo.r + 10 : prm.cg_max_iter;
        const double cg_tol_u = newton ? 1e-12 : prm.cg_tol;
        auto fn = [&](Bin& b, hipStream_t q) {
            const int nus = (int)b.users.size();
            const int cap_pad = host_pow2(b.cap), rsc = b.max_lev + 2;
            const int nchp = geo.nchunk | 1;
            const size_t lds = ustep_small_bytes(geo.ld, b.block, sizeof(T)) + ustep_rows_bytes(b.rcap, nchp) + carve_bytes(b.wcap, 2) +
                               (b.big ? 0 : ustep_big_bytes<T>(b.cap, cap_pad, rsc, 4));
            const size_t bi = (size_t)(&b - &ubins[0]);
            ClusterBufs cb{bar_p + bi * max_clusters, d_xch.p + bi * max_clusters * xch_stride, xch_stride, d_rowcnt.p + bi};
            // clusters: grid <= one workgroup per CU so that every member of every cluster is resident
            const int grid = b.ugrid;
            char* scr = d_scratch.p + (size_t)b.scratch_ofs * scratch_stride;
#define LUS(BL, BG, KK, RS, UN, SY) hipLaunchKernelGGL((k_ustep<T, BL, BG, KK, RS, UN, SY>), dim3(grid), dim3(BL), lds, q, sh, geo, b.d_users.p, nus, d_U.p, d_V.p, prm.lambda, prm.stepsize, cg_max_u, cg_tol_u, strict(), strict(), b.cap, cap_pad, rsc, b.rcap, nchp, scr, scratch_stride, d_counters.p, cb, (tune.fault_cluster_member ? 1 : 0) | (tune.count_rows ? 2 : 0) | (state_of_rejected_V ? 4 : 0), b.wcap, dirp)
#define LU(BL, BG, KK, RS, UN) LUS(BL, BG, KK, RS, UN, 0)
#define LU2(BL) do { if (b.sym) LUS(BL, false, 1, false, 4, 1); else LUS(BL, false, 1, false, 4, 0); } while (0)
#ifndef PCR_NO_OPTIONAL_KERNELS
            if (b.gram) {
                const size_t gl = gram_bytes<T>(b.cap, cap_pad, rsc, geo.ld, nchp, b.block);
                if (b.block == 64)
                    hipLaunchKernelGGL((k_ustep_gram<T, 64>), dim3(nus), dim3(64), gl, q, sh, geo, b.d_users.p, nus, d_U.p, d_V.p, prm.lambda, prm.stepsize,
                                       prm.cg_max_iter, prm.cg_tol, strict(), strict(), b.cap, cap_pad, rsc, nchp, d_counters.p, tune.count_rows);
                else
                    hipLaunchKernelGGL((k_ustep_gram<T, 256>), dim3(nus), dim3(256), gl, q, sh, geo, b.d_users.p, nus, d_U.p, d_V.p, prm.lambda, prm.stepsize,
                                       prm.cg_max_iter, prm.cg_tol, strict(), strict(), b.cap, cap_pad, rsc, nchp, d_counters.p, tune.count_rows);
                return;
            }
#endif
            if (b.big) { if (b.K == 4) LU(512, true, 4, true, 8); else if (b.unr == 8) LU(512, true, 1, true, 8); else LU(512, true, 1, false, 4); }
            else if (b.block == 64) { if (b.rcap > 0) LU(64, false, 1, true, 4); else LU2(64); }
            else if (b.block == 256) LU2(256);
            else if (b.K == 4) LU(512, false, 4, true, 8);
            else if (b.unr == 8) LU(512, false, 1, true, 8);
            else LU2(512);
#undef LU2
#undef LU
#undef LUS
        };
        RC(for_ubins(fn));
        return PCR_OK;
    }

    // pcrpp.cpp:818-838
    // The U step in two halves, so that a training loop can queue the next V step behind it without a host round trip:
    // ustep_launch_async() queues the kernels, the objective sums and the copies of the results into pinned buffers of
    // their own; ustep_finish() reads those buffers once the stream is known to have passed them.
    int update_U(double* now_obj, int64_t* info) override {
        RC(ustep_launch_async());
        HIPCHK(hipStreamSynchronize(st));
        return ustep_finish(now_obj, info);
    }
    bool uobj_merged = false, start_sums_queued = false;
    int ustep_launch_async(bool v_step_follows = false) {
        RC(need_sorted());
        RC(launch_ustep());
        unorm_valid = false;
        // U changed, and k_ustep left the sorted state of (U_new, V) behind: still valid for the next V step -- unless
        // it started from the state of a rejected V_new, which the users it skipped still carry
        if (state_of_rejected_V) { have_sorted = false; state_of_rejected_V = false; }
#ifdef PCR_USTEP_PROF
        HIPCHK(hipMemcpyAsync(h_counters, d_counters.p, (4 + 64) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
#endif
        // sum_i obj_u(i), |V|^2 (:835).  When a V step follows at once from the state this U step left (the pipelined loop),
        // the three sums of ITS starting objective -- the per-user losses, |V|^2, |U|^2 -- ride on the same pass
        // (slots 4..6, where update_V expects them; the U step's own sum in slot 7).
        uobj_merged = v_step_follows && have_sorted;
        if (uobj_merged) { RC(objective_sums(d_objp.p, d_V.p, true, 4, d_objr.p, true)); start_sums_queued = true; }
        else RC(objective_sums(d_objr.p, d_V.p, false, 0, nullptr, true));
        counters_zeroed = true;                                     // k_fin4 has reset them
        HIPCHK(hipMemcpyAsync(h_uobj, d_scal.p, 16 * sizeof(double), hipMemcpyDeviceToHost, st));
        ustep_pending = true;
        return PCR_OK;
    }
    int ustep_finish(double* now_obj, int64_t* info) {
        ustep_pending = false;
        if (p2p && p2p->exchange_failed()) { pcr_set_error("p2p all-reduce: " + p2p->err); return PCR_ERR_COMM; }
#ifdef PCR_USTEP_PROF
        {   // developer build only: per-phase shader-clock totals of thread 0 of every workgroup, by bin class
            static const char* ph[] = {"load", "g.sweep", "g.axpy", "cg.sddmm", "cg.sweep", "cg.axpy", "vec", "ls.sddmm", "ls.sort", "ls.obj", "store", "TOTAL", "wgs"};
            static const char* cl[] = {"64", "256", "512", "cluster"};
            for (int c = 0; c < 4; ++c) {
                const unsigned long long* q = h_counters + 4 + c * 16;
                if (!q[12]) continue;
                fprintf(stderr, "ustep-prof %-8s wgs %5llu  kclk/wg %8.1f :", cl[c], q[12], q[11] / 1000.0 / q[12]);
                for (int i = 0; i < 11; ++i) fprintf(stderr, " %s %.1f%%", ph[i], 100.0 * q[i] / (double)q[11]);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (now_obj) *now_obj = (uobj_merged ? h_uobj[7] : h_uobj[0]) + prm.lambda / 2.0 * (uobj_merged ? h_uobj[5] : h_uobj[1]);      // :835
        const int cb = uobj_merged ? 8 : 4;                         // counters follow the four sums (objective_sums)
        if (h_uobj[cb + 2] != 0.0) { pcr_set_error("k_ustep: a workgroup cluster timed out at a hand-off (members not co-resident?)"); return PCR_ERR_DEVICE; }
        if (info) { info[0] = (int64_t)h_uobj[cb]; info[1] = (int64_t)h_uobj[cb + 1]; }
        ustep_rows += h_uobj[cb + 3];
        return PCR_OK;
    }

    // util.cpp:434-542
    int evaluate(int which, int ndcg_k, double* err, double* ndcg) override {
        if (which != 0 && which != 1) { pcr_set_error("which must be 0 (train) or 1 (test)"); return PCR_ERR_ARG; }
        if (ndcg_k < 1) { pcr_set_error("ndcg_k must be >= 1"); return PCR_ERR_ARG; }
        EvalSet& es = ev[which];
        if (es.idcg_k != ndcg_k) {
            // ideal DCG and discounts with the reference's arithmetic (util.cpp:505-524)
            std::vector<double> idcg(n_users), disc(ndcg_k);
            for (int k = 1; k <= ndcg_k; ++k) disc[k - 1] = 1.0 / log2((double)k + 1.0);
            // (only the ndcg_k largest ratings of a user enter: a partial sort, users side by side on the host threads)
            if (es.max_raw_levels <= 64) {
                // the ndcg_k largest ratings of a user = its levels from the top, each as often as it occurs (equal ratings have
                // equal gains: the order among them does not matter) -- the same terms in the same order as util.cpp:505-524
                pcr_parallel_ranges(n_users, pcr_host_threads(), [&](int, int64_t lo, int64_t hi) {
                    for (int64_t u = lo; u < hi; ++u) {
                        const int64_t o = es.h_runofs[u], Tn = es.h_runofs[u + 1] - o - 1;
                        const int64_t nowk = std::min<int64_t>(ndcg_k, es.h_uptr[u + 1] - es.h_uptr[u]);
                        double m = 0.0;
                        int64_t k = 1;
                        for (int64_t l = Tn - 1; l >= 0 && k <= nowk; --l)
                            for (int32_t c = 0; c < es.h_cnt[o + l] && k <= nowk; ++c, ++k) m += es.h_lgain[o + l] / log2((double)k + 1.0);
                        idcg[u] = m;
                    }
                });
            } else
            pcr_parallel_ranges(n_users, pcr_host_threads(), [&](int, int64_t lo, int64_t hi) {
                std::vector<double> tmp;
                for (int64_t u = lo; u < hi; ++u) {
                    tmp.assign(es.h_val.begin() + es.h_uptr[u], es.h_val.begin() + es.h_uptr[u + 1]);
                    const int64_t nowk = std::min<int64_t>(ndcg_k, (int64_t)tmp.size());
                    std::partial_sort(tmp.begin(), tmp.begin() + nowk, tmp.end(), [](double a, double b) { return a > b; });
                    double m = 0.0;
                    for (int64_t k = 1; k <= nowk; ++k) m += (pow(2.0, tmp[k - 1]) - 1.0) / log2((double)k + 1.0);
                    idcg[u] = m;
                }
            });
            if (n_users) HIPCHK(hipMemcpy(es.idcg.p, idcg.data(), n_users * sizeof(double), hipMemcpyHostToDevice));
            RC(es.disc.upload(disc, st));
            HIPCHK(hipStreamSynchronize(st));
            es.idcg_k = ndcg_k;
        }
        // dcg uses gain/discount products in the reference's order: gain / log2(k+1); keep the division exact
        {
            const bool fast_eval = es.max_raw_levels <= 64;
            const int64_t* up = which == 0 ? d_uptr.p : es.uptr.p;
            const int32_t* it = which == 0 ? d_item.p : es.item.p;
            auto fn = [&](Bin& b, hipStream_t q) {
                const int nus = (int)b.users.size();
                if (fast_eval && b.big) {                // ... users beyond 4096 ratings: the same in a global-scratch slice
                    const int cap_pad = host_pow2(b.cap), rsc = b.max_lev + 2;
                    const size_t lds2 = small_common(512) + carve_bytes(rsc, 4);
                    hipLaunchKernelGGL((k_eval2<T, 512, true>), dim3(std::min(nus, scratch_blocks)), dim3(512), lds2, q, up, it, es.elvl_p, es.erunofs_p,
                                       es.erunstart_p, es.gain.p, es.idcg.p, es.disc.p, ndcg_k, b.d_users.p, nus, d_U.p, d_V.p, geo, d_out4.p, b.cap, cap_pad,
                                       rsc, d_scratch.p, scratch_stride);
                    return;
                }
                if (fast_eval && !b.big) {               // O(len T log len): sort by (raw level, score)
                    const int cap_pad = host_pow2(b.cap), rsc = b.max_lev + 2;
                    const size_t lds2 = small_common(b.block) + eval2_bytes<T>(b.cap, cap_pad, rsc);
#define LE2(BL) hipLaunchKernelGGL((k_eval2<T, BL>), dim3(nus), dim3(BL), lds2, q, up, it, es.elvl_p, es.erunofs_p, es.erunstart_p, es.gain.p, es.idcg.p, es.disc.p, ndcg_k, b.d_users.p, nus, d_U.p, d_V.p, geo, d_out4.p, b.cap, cap_pad, rsc)
                    if (b.block == 64) LE2(64); else if (b.block == 256) LE2(256); else LE2(512);
#undef LE2
                    return;
                }
                const size_t smallb = small_common(b.block) + carve_bytes(b.block / PCR_WAVE + 1, sizeof(T)) + carve_bytes(b.block / PCR_WAVE + 1, 4);
                const size_t lds = smallb + (b.big ? 0 : eval_bytes<T>(b.cap));
                const int grid = b.big ? std::min(nus, scratch_blocks) : nus;
#define LE(BL, BG) hipLaunchKernelGGL((k_eval<T, BL, BG>), dim3(grid), dim3(BL), lds, q, up, it, es.val.p, es.gain.p, es.idcg.p, es.disc.p, ndcg_k, b.d_users.p, nus, d_U.p, d_V.p, geo, d_out4.p, b.cap, d_scratch.p, scratch_stride)
                if (b.big) LE(512, true);
                else if (b.block == 64) LE(64, false);
                else if (b.block == 256) LE(256, false);
                else LE(512, false);
#undef LE
            };
            RC(for_bins(es.bins, "eval", fn));
            const int nb = (int)std::min<int64_t>(512, std::max<int64_t>(1, cdiv(n_users, 2048)));
            const int per = cdiv(std::max<int64_t>(n_users, 1), nb);
            hipLaunchKernelGGL(k_sum4_stage1, dim3(nb), dim3(PCR_EW_BLOCK), 0, st, d_out4.p, n_users, per, d_partA.p);
            hipLaunchKernelGGL(k_fin4, dim3(1), dim3(PCR_EW_BLOCK), 0, st, d_partA.p, nb, d_scal.p + 16);
            HIPCHK(hipGetLastError());
        }
        RC(allreduce_f64(d_scal.p + 16, 4));
        HIPCHK(hipMemcpyAsync(h_scal + 16, d_scal.p + 16, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
        RC(sync_checked());
        if (err) *err = h_scal[16] / h_scal[17];                    // util.cpp:537
        if (ndcg) *ndcg = h_scal[18] / h_scal[19];
        return PCR_OK;
    }

    // pcrpp.cpp:841-901 / pcr.cpp:616-704
    int train(pcr_log_fn log, void* ctx, pcr_iter_stats* hist) override {
        char line[512];
        // (a caller's callback receives the lines on EVERY rank -- they are computed from all-reduced values, the same text
        // everywhere -- so that a job can act on them rank by rank, e.g. deposit its rows for a snapshot; only rank 0 prints)
        auto emit = [&](const char* s) { if (log) log(ctx, s); else if (rank == 0) { fputs(s, stdout); fputc('\n', stdout); fflush(stdout); } };
        const bool pp = prm.solver_type == PCR_SOLVER_PCRPP;
        snprintf(line, sizeof line, "running %s ndcg_k is %d", pp ? "PrimalCR++" : "PrimalCR", prm.ndcg_k); emit(line);
        snprintf(line, sizeof line, "using %d threads. ", prm.threads); emit(line);
        pcr_iter_stats cur;
        memset(&cur, 0, sizeof cur);
        auto do_eval = [&](pcr_iter_stats& rec) -> int {
            if (!prm.do_predict) return PCR_OK;
            RC(evaluate(0, prm.ndcg_k, &rec.train_err, &rec.train_ndcg));
            snprintf(line, sizeof line, "(Training) pairwise error is %g and ndcg is %g", rec.train_err, rec.train_ndcg); emit(line);
            if (tnnz_file != 0) {
                RC(evaluate(1, prm.ndcg_k, &rec.test_err, &rec.test_ndcg));
                snprintf(line, sizeof line, "(Testing) pairwise error is %g and ndcg is %g", rec.test_err, rec.test_ndcg); emit(line);
            }
            return PCR_OK;
        };
        double now_obj = 0.0;
        RC(launch_prepare(d_V.p));                          // :857
        RC(full_objective(d_V.p, &now_obj));                         // :858
        cur.obj = now_obj;
        snprintf(line, sizeof line, "Iter 0 time 0 obj %g", now_obj); emit(line);
        RC(do_eval(cur));
        if (hist) hist[0] = cur;
        if (!prm.do_predict && !log) {
            // no evaluation between iterations and nobody but stdout listening: the pipelined loop (iterate()); an
            // iteration's line is printed once its results are in, i.e. during the next iteration's line search (a caller's
            // log callback may look at the factors when it sees a line -- omp-pmf-train --snapshot-every does -- so with a
            // callback the loop below keeps the factors and the lines in step)
            return iterate(prm.maxiter, hist ? hist + 1 : nullptr, [&](int iter, const pcr_iter_stats& rec) {
                snprintf(line, sizeof line, "Iter %d time %g obj %g", iter, rec.seconds, rec.obj); emit(line);
            });
        }
        double total_time = 0.0;
        for (int iter = 1; iter <= prm.maxiter; ++iter) {
            auto t0 = std::chrono::steady_clock::now();
            int vinfo[3] = {0, 0, 0};
            int64_t uinfo[2] = {0, 0};
            RC(update_V(&now_obj, vinfo));                           // :877
            RC(update_U(&now_obj, uinfo));                           // :878
            HIPCHK(hipStreamSynchronize(st));
            total_time += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            memset(&cur, 0, sizeof cur);
            cur.obj = now_obj; cur.seconds = total_time;
            cur.cg_v = vinfo[0]; cur.ls_v = vinfo[1]; cur.cg_u = uinfo[0]; cur.ls_u = uinfo[1];
            snprintf(line, sizeof line, "Iter %d time %g obj %g", iter, total_time, now_obj); emit(line);
            RC(do_eval(cur));
            if (hist) hist[iter] = cur;
        }
        return PCR_OK;
    }

    // n outer iterations (pcrpp.cpp:869-895 without the evaluation) with ONE host round trip each -- the line search's
    // objective read-back.  The U step of iteration k is queued without waiting for it, the V step of iteration k + 1
    // (gradient, CG, first line-search try) is queued right behind it, and the U step's objective and counters are read
    // when that line search synchronises.  `seconds` is device time (events on the solver's stream), cumulative.
    template <class F>
    int iterate(int n, pcr_iter_stats* out, F on_done) {
        if (n <= 0) return PCR_OK;
        hipEvent_t ev0 = nullptr, evk[2] = {nullptr, nullptr};
        HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&evk[0])); HIPCHK(hipEventCreate(&evk[1]));
        struct Guard { hipEvent_t* e[3]; ~Guard() { for (auto p : e) if (*p) (void)hipEventDestroy(*p); } } guard{{&ev0, &evk[0], &evk[1]}};
        std::vector<pcr_iter_stats> rec(n + 1);
        for (auto& r : rec) memset(&r, 0, sizeof r);
        int rc = PCR_OK;
        device_join = tune.pipeline != 0;
        HIPCHK(hipEventRecord(ev0, st));
        auto close = [&](int k) -> int {                         // iteration k's U step has been finished (fin_*)
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev0, evk[k & 1]));
            rec[k].obj = fin_obj; rec[k].seconds = ms / 1e3; rec[k].cg_u = fin_info[0]; rec[k].ls_u = fin_info[1];
            fin_ready = false;
            if (out) out[k - 1] = rec[k];
            on_done(k, rec[k]);
            return PCR_OK;
        };
        const int pipe = tune.pipeline;
        for (int k = 1; k <= n && rc == PCR_OK; ++k) {
            int vinfo[3] = {0, 0, 0};
            double vobj = 0.0;
            rc = update_V(&vobj, vinfo);                             // finishes the U step of iteration k - 1 on the way
            if (rc != PCR_OK) break;
            if (fin_ready) { rc = close(k - 1); if (rc != PCR_OK) break; }
            rec[k].cg_v = vinfo[0]; rec[k].ls_v = vinfo[1];
            rc = ustep_launch_async(pipe && k < n);
            if (rc != PCR_OK) break;
            HIPCHK(hipEventRecord(evk[k & 1], st));
            if (!pipe) { HIPCHK(hipStreamSynchronize(st)); rc = ustep_finish(&fin_obj, fin_info); if (rc == PCR_OK) rc = close(k); }
        }
        device_join = false;
        if (rc == PCR_OK && ustep_pending) {
            HIPCHK(hipStreamSynchronize(st));
            rc = ustep_finish(&fin_obj, fin_info);
            if (rc == PCR_OK) rc = close(n);
        }
        if (rc != PCR_OK && ustep_pending) { (void)hipStreamSynchronize(st); ustep_pending = false; }
        return rc;
    }
    int iterate_abi(int n, pcr_iter_stats* out) override { return iterate(n, out, [](int, const pcr_iter_stats&) {}); }

    int comm_init(const void* id) override {
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        HIPCHK(hipSetDevice(prm.device));
        NCCLCHK(ncclCommInitRank(&comm, nranks, uid, rank));
        return PCR_OK;
    }
    // direct peer-to-peer exchange (pcr_p2p.h): every rank of the job calls this with the same name
    int comm_init_p2p(const char* shm_name) override {
        if (comm || p2p) { pcr_set_error("this solver already has a communicator"); return PCR_ERR_STATE; }
        HIPCHK(hipSetDevice(prm.device));
        p2p.reset(new P2PComm());
        p2p->ll_timeout_s = std::max(1, tune.p2p_timeout_ms) / 1e3;
        p2p->fault_skip_call = rank == nranks - 1 ? tune.fault_p2p_skip : 0; p2p->fault_coarse = rank == nranks - 1 && tune.fault_p2p_coarse != 0;     // (test hooks: the last rank misbehaves)
        p2p->debug = tune.debug != 0;
        {   // Hardware queues this process holds on its device: the runtime folds the streams of one priority onto at most 4 hardware
            // queues; the null stream (memsets, synchronous copies) is a normal-priority stream, and the runtime keeps a queue of its
            // own for copy kernels.  Published in the control block: ranks that share a GPU share its 24 queue slots (pcr_p2p.h).
            int normal = 2 + (ar_st ? 1 : 0), high = (hi ? 1 : 0) + (int)hi_spare.size();
            for (int i = 0; i < NSIDE; ++i) normal += side[i] ? 1 : 0;
            p2p->my_queues = std::min(4, normal) + std::min(4, high) + 1;
            if (tune.p2p_queue_budget > 0) p2p->queue_budget = tune.p2p_queue_budget;
        }
        if (!p2p->init(shm_name, rank, nranks, (size_t)d2 * geo.ld, sizeof(T), (size_t)std::max(0, tune.p2p_ll) << 20)) {
            pcr_set_error("p2p communicator: " + p2p->err);
            p2p.reset();
            return PCR_ERR_COMM;
        }
        if (p2p->ranks_on_my_device > 1 && nlane > 1) {
            // peers on this GPU (a rehearsal of the N-rank job on fewer GPUs): keep to the solver's stream -- length classes of several
            // ranks side by side on one device's CUs gain nothing, and every lane is a hardware queue the ranks have to share
            nlane = 1;
            if (tune.debug) fprintf(stderr, "[pcr] p2p rank %d: %d ranks on this device -- the U step's classes stay on the solver's stream\n", rank, p2p->ranks_on_my_device);
        }
        if (!p2p->note.empty() && rank == 0) fprintf(stderr, "[pcr] p2p: %s\n", p2p->note.c_str());
        return PCR_OK;
    }
    // a rank that fails leaves the job: its peers must not wait for it
    void comm_abort() override {
        if (p2p) p2p->abort_peers();
        if (comm) { (void)ncclCommAbort(comm); comm = nullptr; }
    }
    int comm_nranks() override {
        if (p2p) return p2p->nranks;
        int n = 1;
        if (comm && ncclCommCount(comm, &n) != ncclSuccess) n = -1;
        return n;
    }
    int sync() override { RC(sync_checked()); return PCR_OK; }
};

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// No C++ exception may cross the C ABI (a host allocation that fails while a 700 M-rating shard is being set up is an error
// code, not std::terminate).
template <class F>
static int abi_guard(const char* what, F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { try { pcr_set_error(std::string(what) + ": out of host memory"); } catch (...) {} return PCR_ERR_NOMEM; }
    catch (const std::exception& e) { try { pcr_set_error(std::string(what) + ": " + e.what()); } catch (...) {} return PCR_ERR_ARG; }
    catch (...) { return PCR_ERR_ARG; }
}
#define PCR_ABI(name, expr) return abi_guard(name, [&]() -> int { return (expr); })

extern "C" {

static int solver_create(const pcr_dataset* ds, const pcr_params* p, int rank, int nranks, int64_t shard_first, int64_t d1_total, pcr_solver** out) {
    if (!ds || !p || !out || nranks < 1 || rank < 0 || rank >= nranks) { pcr_set_error("pcr_solver_create: bad argument"); return PCR_ERR_ARG; }
    if (p->solver_type != PCR_SOLVER_PCR && p->solver_type != PCR_SOLVER_PCRPP) {
        pcr_set_error("wrong solver type (" + std::to_string(p->solver_type) + "): 1 = PrimalCR, 2 = PrimalCR++");
        return PCR_ERR_ARG;
    }
    if (p->precision != PCR_F64 && p->precision != PCR_F32) { pcr_set_error("precision must be PCR_F32 or PCR_F64"); return PCR_ERR_ARG; }
    return abi_guard("pcr_solver_create", [&]() -> int {
        if (p->precision == PCR_F64) {
            std::unique_ptr<Solver<double>> s(new Solver<double>());
            RC(s->init(ds, p, rank, nranks, shard_first, d1_total));
            *out = s.release();
        } else {
            std::unique_ptr<Solver<float>> s(new Solver<float>());
            RC(s->init(ds, p, rank, nranks, shard_first, d1_total));
            *out = s.release();
        }
        return PCR_OK;
    });
}
int pcr_solver_create(const pcr_dataset* ds, const pcr_params* p, int rank, int nranks, pcr_solver** out) {
    return solver_create(ds, p, rank, nranks, -1, 0, out);
}
int pcr_solver_create_shard(const pcr_dataset* ds_local, const pcr_params* p, int rank, int nranks, int64_t first_user, int64_t d1_total,
                            pcr_solver** out) {
    if (first_user < 0) { pcr_set_error("pcr_solver_create_shard: first_user must be >= 0"); return PCR_ERR_ARG; }
    return solver_create(ds_local, p, rank, nranks, first_user, d1_total, out);
}
void pcr_solver_destroy(pcr_solver* s) { delete s; }

// Initialise the HIP runtime for `device` and load this library's code object (the first launch of any of its kernels does):
// a few tenths of a second that a host application can spend on another thread while it is still parsing its input.
int pcr_device_warmup(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pcr_set_error("no HIP device available"); return PCR_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { pcr_set_error("device ordinal out of range"); return PCR_ERR_ARG; }
    return abi_guard("pcr_device_warmup", [&]() -> int {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipFree(nullptr));
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        return PCR_OK;
    });
}

int pcr_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id128) { pcr_set_error("null id"); return PCR_ERR_ARG; }
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return PCR_OK;
}
#define S_OR_ARG if (!s) { pcr_set_error("null solver"); return PCR_ERR_ARG; }
int pcr_solver_comm_init(pcr_solver* s, const void* id128) { S_OR_ARG; PCR_ABI("pcr_solver_comm_init", s->comm_init(id128)); }
int pcr_solver_comm_init_p2p(pcr_solver* s, const char* shm_name) {
    S_OR_ARG;
    if (!shm_name || shm_name[0] != '/') { pcr_set_error("pcr_solver_comm_init_p2p: the name must start with '/' (shm_open)"); return PCR_ERR_ARG; }
    PCR_ABI("pcr_solver_comm_init_p2p", s->comm_init_p2p(shm_name));
}
int pcr_solver_comm_nranks(pcr_solver* s) { if (!s) return -1; return s->comm_nranks(); }
int pcr_solver_setup_phase(const pcr_solver* s, int i, const char** name, double* ms) {
    S_OR_ARG;
    if (i < 0 || i >= (int)s->setup_ms.size()) return PCR_ERR_ARG;      // (the end of the list: no error text)
    if (name) *name = s->setup_ms[(size_t)i].first.c_str();
    if (ms) *ms = s->setup_ms[(size_t)i].second;
    return PCR_OK;
}
int pcr_solver_counter(pcr_solver* s, const char* name, double* value) {
    S_OR_ARG;
    if (!name || !value) { pcr_set_error("pcr_solver_counter: bad argument"); return PCR_ERR_ARG; }
    if (!strcmp(name, "ustep_row_gathers")) { *value = s->ustep_rows; return PCR_OK; }
    if (!strncmp(name, "ustep_row_gathers/", 18)) return s->class_rows(name + 18, value);
    pcr_set_error(std::string("pcr_solver_counter: unknown counter '") + name + "'");
    return PCR_ERR_ARG;
}
// a failing rank tells its peers (p2p: shared error flag; RCCL: local abort) before it reports the error
static int leave_on_error(pcr_solver* s, int rc) { if (rc != PCR_OK) s->comm_abort(); return rc; }
int pcr_solver_set_local_only(pcr_solver* s, int on) { S_OR_ARG; s->local_only = on != 0; return PCR_OK; }
int pcr_solver_shard(const pcr_solver* s, int64_t* first_user, int64_t* n_users, int64_t* nnz_local) {
    S_OR_ARG;
    if (first_user) *first_user = s->first_user;
    if (n_users) *n_users = s->n_users;
    if (nnz_local) *nnz_local = s->nnz_local;
    return PCR_OK;
}
int pcr_solver_set_factors(pcr_solver* s, const double* U, const double* V) { S_OR_ARG; PCR_ABI("pcr_solver_set_factors", s->set_factors(U, V, false)); }
int pcr_solver_get_factors(pcr_solver* s, double* U, double* V) { S_OR_ARG; PCR_ABI("pcr_solver_get_factors", s->get_factors(U, V, false)); }
int pcr_solver_set_factors_local(pcr_solver* s, const double* U_local, const double* V) { S_OR_ARG; PCR_ABI("pcr_solver_set_factors_local", s->set_factors(U_local, V, true)); }
int pcr_solver_get_factors_local(pcr_solver* s, double* U_local, double* V) { S_OR_ARG; PCR_ABI("pcr_solver_get_factors_local", s->get_factors(U_local, V, true)); }
int pcr_comp_m(pcr_solver* s, double* m_out) { S_OR_ARG; PCR_ABI("pcr_comp_m", s->comp_m(m_out)); }
int pcr_objective(pcr_solver* s, double* obj) { S_OR_ARG; PCR_ABI("pcr_objective", s->objective(obj)); }
int pcr_obtain_g(pcr_solver* s, double* g) { S_OR_ARG; PCR_ABI("pcr_obtain_g", s->obtain_g(g)); }
int pcr_compute_Ha(pcr_solver* s, const double* a, double* Ha) { S_OR_ARG; PCR_ABI("pcr_compute_Ha", s->compute_Ha(a, Ha)); }
int pcr_solve_delta(pcr_solver* s, const double* g, double* delta, int* it) { S_OR_ARG; PCR_ABI("pcr_solve_delta", s->solve_delta(g, delta, it)); }
int pcr_update_V(pcr_solver* s, double* now_obj, int* info) { S_OR_ARG; PCR_ABI("pcr_update_V", leave_on_error(s, s->update_V(now_obj, info))); }
int pcr_update_U(pcr_solver* s, double* now_obj, int64_t* info) { S_OR_ARG; PCR_ABI("pcr_update_U", leave_on_error(s, s->update_U(now_obj, info))); }
int pcr_evaluate(pcr_solver* s, int which, int ndcg_k, double* e, double* n) { S_OR_ARG; PCR_ABI("pcr_evaluate", s->evaluate(which, ndcg_k, e, n)); }
int pcr_train(pcr_solver* s, pcr_log_fn log, void* ctx, pcr_iter_stats* hist) { S_OR_ARG; PCR_ABI("pcr_train", leave_on_error(s, s->train(log, ctx, hist))); }
int pcr_solver_sync(pcr_solver* s) { S_OR_ARG; return s->sync(); }
int pcr_iterate(pcr_solver* s, int n, pcr_iter_stats* out) { S_OR_ARG; PCR_ABI("pcr_iterate", leave_on_error(s, s->iterate_abi(n, out))); }

int pcr_profile_enable(pcr_solver* s, int on) { S_OR_ARG; s->prof_on = on != 0; s->prof_period = on > 1 ? on : 1; if (on) s->prof_prewarm(4096); return PCR_OK; }
int pcr_profile_reset(pcr_solver* s) {
    S_OR_ARG;
    s->sync(); s->prof_resolve();
    for (auto& kv : s->prof) { kv.second.ms = 0.0; kv.second.n = 0; kv.second.seen = 0; }
    return PCR_OK;
}
int pcr_solver_ustep_classes(pcr_solver* s, char* buf, int64_t cap) {
    S_OR_ARG;
    const std::string all = s->ustep_classes();
    if (!buf || cap < (int64_t)all.size() + 1) { pcr_set_error("buffer too small"); return PCR_ERR_ARG; }
    memcpy(buf, all.c_str(), all.size() + 1);
    return PCR_OK;
}
int pcr_profile_list(pcr_solver* s, char* buf, int64_t cap) {
    S_OR_ARG;
    std::string all;
    for (auto& kv : s->prof) { if (!all.empty()) all += ","; all += kv.first; }
    if (!buf || cap < (int64_t)all.size() + 1) { pcr_set_error("buffer too small"); return PCR_ERR_ARG; }
    memcpy(buf, all.c_str(), all.size() + 1);
    return PCR_OK;
}
int pcr_profile_get(pcr_solver* s, const char* name, double* total_ms, int64_t* launches) {
    S_OR_ARG;
    s->sync(); s->prof_resolve();
    auto it = s->prof.find(name ? name : "");
    if (total_ms) *total_ms = it == s->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == s->prof.end() ? 0 : it->second.n;
    return PCR_OK;
}

int pcr_profile_launches(pcr_solver* s, const char* name, int64_t* launches) {
    S_OR_ARG;
    auto it = s->prof.find(name ? name : "");
    if (launches) *launches = it == s->prof.end() ? 0 : it->second.seen;
    return PCR_OK;
}

int pcr_profile_scope(pcr_solver* s, const char* name, int64_t* ratings, int64_t* users) {
    S_OR_ARG;
    auto it = s->prof.find(name ? name : "");
    const bool whole = it == s->prof.end() || it->second.ratings < 0;
    if (ratings) *ratings = whole ? s->nnz_local : it->second.ratings;
    if (users) *users = whole ? s->n_users : it->second.users;
    return PCR_OK;
}

int pcr_predict(const double* U, int64_t d1, const double* V, int64_t d2, int64_t k, int64_t n,
                const int32_t* user, const int32_t* item, double* pred, int device) {
    if (!U || !V || k < 1 || n < 0 || (n > 0 && (!user || !item || !pred))) { pcr_set_error("pcr_predict: bad argument"); return PCR_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pcr_set_error("no HIP device available"); return PCR_ERR_DEVICE; }
    return abi_guard("pcr_predict", [&]() -> int {
    HIPCHK(hipSetDevice(device));
    {   // ids inside the model (the reference reads out of bounds on a bad line, pmf-predict.cpp:58): checked by the host threads
        const int nth = pcr_host_threads();
        std::vector<int64_t> bad((size_t)nth, -1);
        pcr_parallel_ranges(n, nth, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t z = lo; z < hi; ++z)
                if (user[z] < 0 || user[z] >= d1 || item[z] < 0 || item[z] >= d2) { bad[(size_t)t] = z; return; }
        });
        int64_t first_bad = -1;
        for (int64_t x : bad) if (x >= 0 && (first_bad < 0 || x < first_bad)) first_bad = x;
        if (first_bad >= 0) { pcr_set_error("pair " + std::to_string(first_bad) + " outside the model"); return PCR_ERR_ARG; }
    }
    Geo geo;
    geo.r = (int)k; geo.ld = ((int)k + 3) & ~3; geo.nchunk = geo.ld / 2; geo.G = std::min(64, host_pow2(geo.nchunk));
    // the model file holds fp64 factors: score in fp64 like pmf-predict.cpp:58-62.  The matrices go up as they are (slabs of 64 M
    // values) and are padded to ld on the device; the ids from the caller's arrays.
    DBuf<double> dU, dV, dP, stage;
    DBuf<int32_t> du, di;
    hipStream_t st = nullptr;
    RC(dU.alloc((size_t)d1 * geo.ld)); RC(dV.alloc((size_t)d2 * geo.ld)); RC(dP.alloc((size_t)n));
    const int64_t slab_rows = std::max<int64_t>(1, ((int64_t)64 << 20) / std::max<int64_t>(1, k));
    RC(stage.alloc((size_t)std::min<int64_t>(std::max(d1, d2), slab_rows) * (size_t)k));
    for (int w = 0; w < 2; ++w) {
        const double* H = w == 0 ? U : V;
        double* D = w == 0 ? dU.p : dV.p;
        const int64_t rows = w == 0 ? d1 : d2;
        for (int64_t r0 = 0; r0 < rows; r0 += slab_rows) {
            const int64_t nr = std::min(slab_rows, rows - r0);
            HIPCHK(hipMemcpyAsync(stage.p, H + r0 * k, (size_t)nr * k * sizeof(double), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((k_mat_in<double>), dim3((unsigned)std::min<int64_t>(1 << 16, cdiv(nr * geo.ld, 256))), dim3(256), 0, st, stage.p, D + r0 * geo.ld, nr, (int)k, geo.ld);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    RC(du.upload_n(user, (size_t)n)); RC(di.upload_n(item, (size_t)n));
    if (n > 0) {
        const int gpb = 256 / geo.G;
        hipLaunchKernelGGL((k_predict<double>), dim3(cdiv(n, gpb)), dim3(256), 0, st, dU.p, dV.p, du.p, di.p, n, geo, dP.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(pred, dP.p, n * sizeof(double), hipMemcpyDeviceToHost));
    }
    return PCR_OK;
    });
}

}  // extern "C"
