// pcr_p2p.h -- direct peer-to-peer all-reduce for the replicated V-side vectors ("direct-P2P
// reduce-scatter / all-gather variant"), the alternative to ncclAllReduce behind pcr_solver_comm_init_p2p().
//
// One process per GPU on ONE node.  Every rank exposes an exchange buffer through a HIP IPC handle; the ranks meet in a
// small POSIX shared-memory control block (handles, a generation barrier, an error flag).  An all-reduce is
//     partial -> own X[parity]                       (device copy)
//     host barrier                                    (stream sync + shm barrier: kernel-boundary visibility across GPUs)
//     reduce-scatter: rank q sums slice q of every peer's X[parity] IN RANK ORDER into its Y[parity]   (reads over xGMI)
//     host barrier
//     all-gather: every rank copies every slice from its owner's Y[parity]                              (reads over xGMI)
// Every rank obtains bitwise-identical sums (fixed order), which the replicated CG recurrence relies on.  Messages of at most
// 256 KB (the 8 objective scalars; small item tables) skip the second phase: every rank sums all of every peer's X itself.
// X and Y are double-buffered by call parity, so a rank that runs ahead never overwrites bytes a peer still reads (a rank
// reaches its next barrier only after its own reads have completed).
//
// Why host barriers there: xGMI peers see each other's plain stores reliably at kernel boundaries; on 16-109 MB vectors two
// ~10 us host round trips per all-reduce are noise, and a rank that dies or times out raises the shared error flag, which every
// peer polls in its barrier: no rank waits forever on a lost peer (with RCCL the surviving ranks block inside the collective).
//
// Small vectors (at most ll_max bytes: the ml1m shape's 1.6 MB, the objective's scalars) take a DEVICE-DRIVEN exchange instead
// -- no staging copy, no stream synchronisation, no host barrier -- in ONE kernel per rank (k_p2p_ll):
//     scatter    rank r stores element i of its partial, packed with the call's sequence number into one 8-byte word (value,
//                seq), straight into inbox[r] of the element's owner q = i / per            (remote 8-byte stores)
//     reduce     the owner polls its inbox words until they carry this call's seq (its own memory: local polls), sums the ranks'
//                values in rank order, and stores (sum, seq) into the outbox of every rank   (remote 8-byte stores)
//     gather     every rank polls its own outbox and unpacks
// Data and flag travel in the SAME 8-byte store (the hardware keeps an aligned 8-byte store whole), so no ordering between a
// payload and its flag is ever assumed -- the scheme NCCL / RCCL call LL.  A slot is reused by the next call only after its
// reader has consumed it (a rank leaves a call only when every owner has answered, and an owner answers only after it read
// every rank's word), so single buffers suffice; seq makes a stale word unmistakable.  fp64 elements travel as two words.
// Waits are bounded by a WALL-CLOCK deadline (wall_clock64, ll_timeout_s: 20 s by default -- ranks may enter their first
// exchange seconds apart, e.g. behind their share of the initial() stream -- pcr_tune "p2p_timeout_ms").  A thread whose
// wait times out raises an error word in pinned host memory and is DEAD for the rest of the kernel: it waits for nothing more
// (every later word returns at once), and in place of a sum it publishes a POISON word (the call's seq with the top bit set),
// at which every peer's waiting thread raises its own error and dies the same way -- no rank consumes a made-up sum as a good
// result, and no rank sits out its own full deadline behind a peer that already knows.  The host reads the error word at its
// next synchronisation, raises the job's shared error flag (so that peers on the host-synchronised path leave too) and
// returns PCR_ERR_COMM.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>

#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define PCR_P2P_MAXR 16

#define PCR_P2P_MAGIC 0x50435250325036ull     // "PCRP2P6"
// Hardware queues one device maps at once for all its processes (gfx950 under the kernel driver's firmware scheduler: 4 pipes x
// 8 queues of the first compute micro-engine, 8 of them the kernel driver's own).  Past it the scheduler time-slices the queues
// -- waves are saved and restored -- and kernels that WAIT FOR EACH OTHER (k_p2p_ll) advance one phase per rotation instead of
// per microsecond: tools/ubench/queue_budget_probe.hip, NOTES.md round 6.  Ranks that share a device publish what they hold; a
// device over this budget sends every rank to the host-synchronised exchange, which needs no two kernels resident together.
#define PCR_P2P_QUEUE_BUDGET 24
// ... of which this many are left to whoever else is on the device and cannot be seen from here (a host application's other HIP
// contexts, a test runner's: their idle queues take slots too)
#define PCR_P2P_QUEUE_RESERVE 8
struct P2PCtl {                               // in POSIX shared memory, created (O_EXCL) and zero-filled by rank 0
    std::atomic<uint64_t> magic;              // written LAST by rank 0: the block is ready
    uint64_t created_ns;                      // CLOCK_REALTIME at creation: a block older than the rendezvous time-out is a dead job's
    uint32_t nranks_expected;
    std::atomic<uint32_t> count, gen, attached;
    std::atomic<int32_t> error;
    std::atomic<uint32_t> posted[PCR_P2P_MAXR];
    hipIpcMemHandle_t handle[PCR_P2P_MAXR];
    uint64_t bytes[PCR_P2P_MAXR];
    hipIpcMemHandle_t ll_handle[PCR_P2P_MAXR];        // the boxes of the device-driven exchange (an allocation of their own)
    uint64_t ll_bytes[PCR_P2P_MAXR];
    uint32_t ll_ok[PCR_P2P_MAXR];                     // 1 = that rank holds FINE-GRAINED boxes (written before posted[] is raised)
    uint32_t queues[PCR_P2P_MAXR];                    // hardware queues that rank's process holds on its device (before posted[] too)
    char bus[PCR_P2P_MAXR][24];                       // PCI address of that rank's device: equal strings = one GPU shared
};

template <typename X> struct P2PPtrs { const X* p[PCR_P2P_MAXR]; };

// out[i - lo] = sum_r src.p[r][i] for i in [lo, hi), ranks in order
template <typename X>
__global__ __launch_bounds__(256) void k_p2p_reduce(X* __restrict__ out, P2PPtrs<X> src, int nranks, int64_t lo, int64_t hi) {
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
        // exchanged bytes are read and written at SYSTEM scope (sc0 sc1: past this GPU's L2, which may still hold the lines
        // of the previous exchange of the same parity, and written through for the peers that read the result)
        X s = __hip_atomic_load(&src.p[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < nranks; ++r) s += __hip_atomic_load(&src.p[r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out[i - lo], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// out[i] = slice owner's y[i - lo_owner]; slices are [q * per, min(n, (q + 1) * per))
template <typename X>
__global__ __launch_bounds__(256) void k_p2p_gather(X* __restrict__ out, P2PPtrs<X> y, int64_t n, int64_t per) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t q = i / per;
        out[i] = __hip_atomic_load(&y.p[q][i - q * per], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- device-driven exchange for small vectors (see the header comment)
struct P2PLLPtrs { unsigned long long* inbox[PCR_P2P_MAXR]; unsigned long long* outbox[PCR_P2P_MAXR]; };
template <typename X> struct LLWords;
template <> struct LLWords<float> { static constexpr int W = 1; };
template <> struct LLWords<double> { static constexpr int W = 2; };
__device__ __forceinline__ void ll_put(unsigned long long* dst, float v, unsigned seq) {
    __hip_atomic_store(dst, ((unsigned long long)seq << 32) | (unsigned)__float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ll_put(unsigned long long* dst, double v, unsigned seq) {
    __hip_atomic_store(dst, ((unsigned long long)seq << 32) | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, ((unsigned long long)seq << 32) | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#define PCR_LL_POISON 0x80000000u              // top bit of a word's tag: "the rank that owed this word gave up" (seq stays below it)
// per thread: wall_clock64 deadline of the kernel; dead = it has failed; phase / peer / idx = what it is waiting for (1 = rank
// `peer`'s contribution to element idx of this rank's slice, 2 = owner `peer`'s answer for element idx)
struct LLWait { long long deadline; bool dead; int phase, peer; long long idx; };
// err[0] = raised; err[1..3] = phase, peer and element of the FIRST wait of this rank that failed; phase 3 = it met a poison word
// (the peer had given up first).  Pinned host memory, written at system scope; the detail is a diagnosis, not a protocol word: two
// threads failing in the same microsecond may both write it.
#define PCR_LL_ERR_WORDS 4
// one word carrying this call's seq -- or 0 with the thread dead and *err raised: the deadline passed, the word is poisoned, or
// another thread of this rank has already failed.  A dead thread returns at once.
__device__ __forceinline__ unsigned ll_word(const unsigned long long* src, unsigned seq, int* err, LLWait& w) {
    if (w.dead) return 0u;
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned tag = (unsigned)(x >> 32);
        if (tag == seq) return (unsigned)x;
        const bool poison = tag == (seq | PCR_LL_POISON);
        // (every 4096 polls, ~0.4 ms: the error word lives in pinned HOST memory -- thousands of waiting threads must not read it often)
        if (poison || ((spins & 4095u) == 0 && ((long long)wall_clock64() - w.deadline > 0 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0))) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {          // the first failure of this rank says what it waited for
                __hip_atomic_store(err + 1, poison ? 3 : w.phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(err + 2, w.peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(err + 3, (int)(w.idx & 0x7fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            w.dead = true;
            return 0u;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ float ll_get(const unsigned long long* src, unsigned seq, int* err, LLWait& w, float) {
    return __int_as_float((int)ll_word(src, seq, err, w));
}
__device__ __forceinline__ double ll_get(const unsigned long long* src, unsigned seq, int* err, LLWait& w, double) {
    const unsigned lo = ll_word(src, seq, err, w), hi = ll_word(src + 1, seq, err, w);
    return __hiloint2double((int)hi, (int)lo);
}
// buf[0, n) <- sum over the ranks, in rank order, the same bits on every rank.  The grid must be co-resident (the host keeps it
// at <= 128 workgroups); inbox of rank q: [sender][per * W] words, outbox: [n_max * W] words.
template <typename X>
__global__ __launch_bounds__(256) void k_p2p_ll(X* __restrict__ buf, int64_t n, int64_t per, int64_t box_stride, int me, int nranks,
                                                P2PLLPtrs ll, unsigned seq, int* err, long long budget_ticks) {
    constexpr int W = LLWords<X>::W;
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, dt = (int64_t)gridDim.x * 256;
    LLWait wt{(long long)wall_clock64() + budget_ticks, false, 0, 0, 0};
    for (int64_t i = t0; i < n; i += dt) {                                           // scatter
        const int64_t q = i / per;
        ll_put(ll.inbox[q] + ((int64_t)me * box_stride + (i - q * per)) * W, buf[i], seq);
    }
    const int64_t lo = per * me < n ? per * me : n, hi = lo + per < n ? lo + per : n;
    for (int64_t j = lo + t0; j < hi; j += dt) {                                     // reduce my slice, answer everybody
        wt.phase = 1; wt.peer = 0; wt.idx = j;
        X s = ll_get(ll.inbox[me] + (j - lo) * W, seq, err, wt, X());
        for (int r = 1; r < nranks; ++r) { wt.peer = r; s += ll_get(ll.inbox[me] + ((int64_t)r * box_stride + (j - lo)) * W, seq, err, wt, X()); }
        // (a dead thread has no sum: it answers with poison, so that every rank fails this call instead of consuming garbage)
        for (int p = 0; p < nranks; ++p) ll_put(ll.outbox[p] + j * W, wt.dead ? X(0) : s, wt.dead ? (seq | PCR_LL_POISON) : seq);
    }
    for (int64_t i = t0; i < n; i += dt) {                                           // gather
        wt.phase = 2; wt.peer = (int)(i / per); wt.idx = i;
        buf[i] = ll_get(ll.outbox[me] + i * W, seq, err, wt, X());
    }
}

struct P2PComm {
    int rank = 0, nranks = 1;
    P2PCtl* ctl = nullptr;
    std::string shm_name;
    char* xbuf = nullptr;                     // this rank's exchange buffer: X[2][cap] | Y[2][slice_cap] | S[2][64 doubles]
    char* peer[PCR_P2P_MAXR] = {};
    size_t cap_bytes = 0, slice_bytes = 0;    // one X / one Y (bytes)
    uint64_t seq = 0;
    double timeout_s = 120.0;                 // host barriers and the rendezvous
    double ll_timeout_s = 20.0;               // wall-clock deadline of one device-driven exchange (pcr_tune "p2p_timeout_ms")
    long long ll_ticks = 0;                   // ... in wall_clock64 ticks of this device
    int fault_skip_call = 0;                  // test hook (pcr_tune "fault_p2p_skip"): this rank never launches its n-th device-driven exchange
    bool fault_coarse = false;                // test hook (pcr_tune "fault_p2p_coarse"): pretend the fine-grained allocation failed on this rank
    bool debug = false;
    int ll_calls = 0;
    // what this rank's process holds on its device, for the budget ranks that share a device must keep (PCR_P2P_QUEUE_BUDGET)
    int my_queues = 1;
    int queue_budget = PCR_P2P_QUEUE_BUDGET;
    int ranks_on_my_device = 1, queues_on_my_device = 1;
    bool ll_off_by_budget = false;            // the device-driven exchange was switched off for the whole job by that budget
    std::string note;                         // ... and why, in words (pcr_solver_comm_init_p2p prints it once, on rank 0)
    std::string err;
    // device-driven exchange (k_p2p_ll): two box sets in this rank's buffer, one for vectors of at most ll_elems elements of elt
    // bytes, one for the scalars; per-set sequence numbers; an error word the kernels raise, in pinned host memory
    size_t ll_max_bytes = 0, ll_elems = 0, ll_per = 0, ll_elt = 4;
    size_t ll_off[2] = {0, 0}, ll_outofs[2] = {0, 0};         // byte offsets of a set's inbox / its outbox inside a rank's buffer
    unsigned ll_seq[2] = {0, 0};
    int* ll_err = nullptr;
    static constexpr size_t LL_SCAL = 64;                     // doubles in the scalar set
    // The boxes are FINE-GRAINED device memory (hipExtMallocWithFlags): words a peer stores over xGMI while this rank's kernel is
    // polling must not be hidden behind lines this GPU's L2 holds for its own (coarse-grained) memory -- the reason RCCL keeps its
    // LL buffers in such memory too.  (On one device every rank goes through the same L2, which is how the tests run.)
    char* llbuf = nullptr;
    char* ll_peer[PCR_P2P_MAXR] = {};

    static constexpr size_t SCAL_BYTES = 64 * sizeof(double);
    size_t ll_bytes() const { return ll_outofs[1] ? ll_outofs[1] + LL_SCAL * 16 : 0; }
    size_t total_bytes() const { return 2 * cap_bytes + 2 * slice_bytes + 2 * SCAL_BYTES; }
    char* X_of(char* base, int par) const { return base + (size_t)par * cap_bytes; }
    char* Y_of(char* base, int par) const { return base + 2 * cap_bytes + (size_t)par * slice_bytes; }
    char* S_of(char* base, int par) const { return base + 2 * cap_bytes + 2 * slice_bytes + (size_t)par * SCAL_BYTES; }

    bool fail(const std::string& m) { err = m; if (ctl) ctl->error.store(1); return false; }

    // host barrier over the control block; false on a peer's error or a time-out
    bool barrier() {
        const uint32_t g = ctl->gen.load(std::memory_order_acquire);
        if (ctl->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
            ctl->count.store(0, std::memory_order_relaxed);
            ctl->gen.fetch_add(1, std::memory_order_release);
            return ctl->error.load() == 0 || fail("a peer rank reported an error");
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; ctl->gen.load(std::memory_order_acquire) == g; ++spins) {
            if (ctl->error.load(std::memory_order_relaxed)) return fail("a peer rank reported an error");
            if ((spins & 1023u) == 1023u) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                    return fail("timed out waiting for the peer ranks");
                std::this_thread::yield();
            }
        }
        return ctl->error.load() == 0 || fail("a peer rank reported an error");
    }

    static uint64_t now_ns() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
    // elems_max: longest vector (elements of elt bytes) that will be all-reduced
    // Rendezvous: rank 0 removes whatever carries the name (a crashed job's segment still holds posted[] = 1, dead IPC handles,
    // maybe error = 1), creates the block with O_CREAT | O_EXCL, fills it and writes the magic word LAST; the other ranks open
    // the name until they find a block whose magic, rank count and age say it is THIS job's.  Callers should put something
    // unpredictable into the name (the CLI and bench.py do).
    bool init(const char* name, int rank_, int nranks_, size_t elems_max, size_t elt, size_t ll_max = 0) {
        rank = rank_; nranks = nranks_; ll_max_bytes = ll_max; ll_elt = elt;
        if (nranks > PCR_P2P_MAXR) return fail("p2p communicator: at most 16 ranks");
        shm_name = name;
        const auto t_open = std::chrono::steady_clock::now();
        auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count(); };
        if (rank == 0) {
            shm_unlink(name);
            const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return fail(std::string("shm_open ") + name + ": " + strerror(errno));
            if (ftruncate(fd, sizeof(P2PCtl)) != 0) { close(fd); shm_unlink(name); return fail("ftruncate on the control block failed"); }
            void* m = mmap(nullptr, sizeof(P2PCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) { shm_unlink(name); return fail("mmap of the control block failed"); }
            ctl = static_cast<P2PCtl*>(m);
            ctl->created_ns = now_ns();
            ctl->nranks_expected = (uint32_t)nranks;
            ctl->magic.store(PCR_P2P_MAGIC, std::memory_order_release);
        } else {
            for (;;) {
                const int fd = shm_open(name, O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat sb;
                    void* m = (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(P2PCtl))
                                  ? mmap(nullptr, sizeof(P2PCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
                    close(fd);
                    if (m != MAP_FAILED) {
                        P2PCtl* c = static_cast<P2PCtl*>(m);
                        // (the magic word first: rank 0 writes it LAST, with release -- the plain fields may only be read
                        // behind this acquire; found by the ThreadSanitizer harness, sanitize/p2p_tsan_harness.cpp)
                        if (c->magic.load(std::memory_order_acquire) == PCR_P2P_MAGIC && c->nranks_expected == (uint32_t)nranks &&
                            now_ns() - c->created_ns < (uint64_t)(timeout_s * 1e9)) { ctl = c; break; }
                        munmap(m, sizeof(P2PCtl));               // not (yet) this job's block: rank 0 will replace it
                    }
                }
                if (waited() > timeout_s) { err = "no control block from rank 0 (rendezvous timed out)"; return false; }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        cap_bytes = ((elems_max * elt) + 255) & ~(size_t)255;
        const size_t per = (elems_max + nranks - 1) / nranks;
        slice_bytes = ((per * elt) + 255) & ~(size_t)255;
        if (ll_max_bytes) {       // box sets behind X | Y | S: [inbox: nranks x per x W words][outbox: n x W words], vectors then scalars
            const size_t W = elt / 4;
            ll_elems = std::min(elems_max, ll_max_bytes / elt);
            ll_per = (ll_elems + nranks - 1) / nranks;
            const size_t sper = (LL_SCAL + nranks - 1) / nranks;
            ll_off[0] = 0;
            ll_outofs[0] = ll_off[0] + (size_t)nranks * ll_per * W * 8;
            ll_off[1] = (ll_outofs[0] + ll_elems * W * 8 + 255) & ~(size_t)255;
            ll_outofs[1] = ll_off[1] + (size_t)nranks * sper * 2 * 8;
            if (hipHostMalloc((void**)&ll_err, PCR_LL_ERR_WORDS * sizeof(int)) != hipSuccess) return fail("hipHostMalloc of the exchange's error word failed");
            for (int w = 0; w < PCR_LL_ERR_WORDS; ++w) ll_err[w] = 0;
        }
        if (hipMalloc((void**)&xbuf, total_bytes()) != hipSuccess) return fail("hipMalloc of the exchange buffer failed");
        if (hipMemset(xbuf, 0, total_bytes()) != hipSuccess) return fail("hipMemset of the exchange buffer failed");
        ctl->ll_ok[rank] = 0;
        if (ll_max_bytes) {
            // The boxes must be fine-grained: a kernel that polls coarse-grained memory may never see a peer's store over xGMI
            // (it can stay hidden behind a line this GPU's L2 holds), so every poll would run into its deadline.  A rank that
            // cannot get such memory says so in the control block, and then EVERY rank takes the host-synchronised path.
            if (fault_coarse || hipExtMallocWithFlags((void**)&llbuf, ll_bytes(), hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                llbuf = nullptr;
                if (debug) fprintf(stderr, "[pcr] p2p rank %d: no fine-grained memory for the exchange boxes -- device-driven exchange off\n", rank);
            } else {
                if (hipMemset(llbuf, 0, ll_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail("hipMemset of the exchange boxes failed");
                if (hipIpcGetMemHandle(&ctl->ll_handle[rank], llbuf) != hipSuccess) return fail("hipIpcGetMemHandle of the exchange boxes failed");
                ctl->ll_bytes[rank] = ll_bytes();
                ctl->ll_ok[rank] = 1;
            }
            int khz = 0, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
            ll_ticks = (long long)(ll_timeout_s * 1e3 * (double)khz);
        }
        // (dmabuf IPC: on hosts whose driver has no legacy IPC mode the process must run with HSA_ENABLE_IPC_MODE_LEGACY=0 --
        // an environment prerequisite of the ROCm runtime, listed in include/primalcr.h; the library itself reads no variable)
        if (hipIpcGetMemHandle(&ctl->handle[rank], xbuf) != hipSuccess) return fail("hipIpcGetMemHandle failed (on dmabuf-only hosts run with HSA_ENABLE_IPC_MODE_LEGACY=0)");
        ctl->bytes[rank] = total_bytes();
        {   // which GPU this rank is on and what it holds there (ranks that share a GPU share its hardware queues)
            char bus[24] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess || !bus[0]) snprintf(bus, sizeof bus, "rank-%d", rank);
            bus[sizeof bus - 1] = 0;
            memcpy(ctl->bus[rank], bus, sizeof bus);
            ctl->queues[rank] = (uint32_t)std::max(1, my_queues);
        }
        ctl->posted[rank].store(1, std::memory_order_release);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < nranks; ++r)
            while (!ctl->posted[r].load(std::memory_order_acquire)) {
                if (ctl->error.load()) return fail("a peer rank reported an error during the rendezvous");
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("rendezvous timed out");
                std::this_thread::yield();
            }
        {   // (posted[] was raised after bus[] / queues[]: every rank computes the same sums from the same words)
            int worst_n = 1, worst_q = 0, worst_r = rank;
            for (int a = 0; a < nranks; ++a) {
                int n_same = 0, q_sum = 0;
                for (int b = 0; b < nranks; ++b)
                    if (strncmp(ctl->bus[a], ctl->bus[b], sizeof ctl->bus[a]) == 0) { ++n_same; q_sum += (int)ctl->queues[b]; }
                if (a == rank) { ranks_on_my_device = n_same; queues_on_my_device = q_sum; }
                if (n_same > 1 && q_sum > worst_q) { worst_n = n_same; worst_q = q_sum; worst_r = a; }
            }
            if (worst_n > 1 && worst_q > queue_budget - PCR_P2P_QUEUE_RESERVE && ll_max_bytes) {
                ll_max_bytes = 0;
                ll_off_by_budget = true;
                char bus[25] = {};
                memcpy(bus, ctl->bus[worst_r], 24);
                note = std::to_string(worst_n) + " ranks share device " + bus + " and hold " + std::to_string(worst_q) + " hardware queues on it together (the device maps " +
                       std::to_string(queue_budget) + " at once, " + std::to_string(PCR_P2P_QUEUE_RESERVE) + " are left to other processes): kernels that wait for each other are not guaranteed to run side by side there -- every exchange of this job is host-synchronised";
            }
        }
        if (ll_max_bytes) {       // posted[] was raised after ll_ok[]: every rank takes the same decision from the same words
            bool all_ok = true;
            for (int r = 0; r < nranks; ++r) all_ok = all_ok && ctl->ll_ok[r] == 1;
            if (!all_ok) {
                ll_max_bytes = 0;
                if (debug && rank == 0) fprintf(stderr, "[pcr] p2p: a rank has no fine-grained exchange boxes: all ranks take the host-synchronised exchange\n");
            }
        }
        if (debug && rank == 0)
            fprintf(stderr, "[pcr] p2p: %d ranks, vectors up to %zu bytes %s\n", nranks, ll_max_bytes ? ll_elems * elt : (size_t)0,
                    ll_max_bytes ? "device-driven (fine-grained boxes), larger ones host-synchronised" : "-- every exchange host-synchronised");
        for (int r = 0; r < nranks; ++r) {
            if (r == rank) { peer[r] = xbuf; continue; }
            if (ctl->bytes[r] != total_bytes()) return fail("ranks disagree on the exchange buffer size");
            void* p = nullptr;
            if (hipIpcOpenMemHandle(&p, ctl->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                return fail("hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s buffer failed");
            peer[r] = static_cast<char*>(p);
            if (ll_max_bytes) {
                if (ctl->ll_bytes[r] != ll_bytes()) return fail("ranks disagree on the size of the exchange boxes");
                void* q = nullptr;
                if (hipIpcOpenMemHandle(&q, ctl->ll_handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                    return fail("hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s exchange boxes failed");
                ll_peer[r] = static_cast<char*>(q);
            }
        }
        ll_peer[rank] = llbuf;
        ctl->attached.fetch_add(1);
        if (!barrier()) return false;
        if (rank == 0) shm_unlink(name);      // everyone is attached: the name can go (the mapping lives on)
        ready = true;
        return true;
    }

    // in-place sum of buf[0, n) over the ranks; stream-ordered on st from the caller's point of view
    template <typename X>
    bool allreduce(X* buf, size_t n, hipStream_t st, bool scalars = false) {
        if (ll_err && *ll_err) return fail(timeout_text());
        if (ll_max_bytes && n > 0 && (scalars ? (sizeof(X) == 8 && n <= LL_SCAL) : (sizeof(X) == ll_elt && n <= ll_elems))) {
            // device-driven: one kernel, stream-ordered, nothing on the host
            const int set = scalars ? 1 : 0;
            const size_t per = scalars ? (LL_SCAL + nranks - 1) / nranks : ll_per;
            P2PLLPtrs ll;
            for (int r = 0; r < nranks; ++r) {
                ll.inbox[r] = reinterpret_cast<unsigned long long*>(ll_peer[r] + ll_off[set]);
                ll.outbox[r] = reinterpret_cast<unsigned long long*>(ll_peer[r] + ll_outofs[set]);
            }
            unsigned sq = ++ll_seq[set] & (PCR_LL_POISON - 1);            // tags 1 .. 2^31 - 1 (0 = an untouched box, top bit = poison)
            if (sq == 0) sq = ++ll_seq[set] & (PCR_LL_POISON - 1);
            const int grid = (int)std::min<size_t>(128, (n + 255) / 256);
            if (fault_skip_call && ++ll_calls == fault_skip_call) return true;       // test hook: this rank "forgets" one exchange
            hipLaunchKernelGGL((k_p2p_ll<X>), dim3(grid), dim3(256), 0, st, buf, (int64_t)n, (int64_t)per, (int64_t)per, rank, nranks, ll, sq, ll_err,
                               ll_ticks);
            return hipGetLastError() == hipSuccess || fail("p2p: exchange launch failed");
        }
        const int par = (int)(seq++ & 1);
        const size_t bytes = n * sizeof(X);
        if (!scalars && bytes > cap_bytes) return fail("p2p all-reduce larger than the exchange buffer");
        if (scalars && bytes > SCAL_BYTES) return fail("p2p scalar all-reduce larger than its slot");
        auto mine = [&](char* base) { return scalars ? S_of(base, par) : X_of(base, par); };
        if (hipMemcpyAsync(mine(xbuf), buf, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("p2p: staging copy failed");
        if (hipStreamSynchronize(st) != hipSuccess) return fail("p2p: stream error before the exchange");
        if (!barrier()) return false;
        P2PPtrs<X> src;
        for (int r = 0; r < nranks; ++r) src.p[r] = reinterpret_cast<const X*>(mine(peer[r]));
        const bool one_shot = scalars || bytes <= ((size_t)256 << 10);
        if (one_shot) {
            const int grid = (int)std::min<size_t>(1024, (n + 255) / 256);
            hipLaunchKernelGGL((k_p2p_reduce<X>), dim3(grid), dim3(256), 0, st, buf, src, nranks, (int64_t)0, (int64_t)n);
            return hipGetLastError() == hipSuccess || fail("p2p: reduce launch failed");
        }
        const int64_t per = (int64_t)((n + nranks - 1) / nranks);
        const int64_t lo = std::min<int64_t>((int64_t)n, per * rank), hi = std::min<int64_t>((int64_t)n, lo + per);
        if (hi > lo) {
            const int grid = (int)std::min<int64_t>(2048, (hi - lo + 255) / 256);
            hipLaunchKernelGGL((k_p2p_reduce<X>), dim3(grid), dim3(256), 0, st, reinterpret_cast<X*>(Y_of(xbuf, par)), src, nranks, lo, hi);
        }
        if (hipStreamSynchronize(st) != hipSuccess) return fail("p2p: stream error in the reduce-scatter");
        if (!barrier()) return false;
        P2PPtrs<X> ys;
        for (int r = 0; r < nranks; ++r) ys.p[r] = reinterpret_cast<const X*>(Y_of(peer[r], par));
        const int grid = (int)std::min<size_t>(2048, (n + 255) / 256);
        hipLaunchKernelGGL((k_p2p_gather<X>), dim3(grid), dim3(256), 0, st, buf, ys, (int64_t)n, per);
        return hipGetLastError() == hipSuccess || fail("p2p: gather launch failed");
    }

    void abort_peers() { if (ctl) ctl->error.store(1); }
    // has a device-driven exchange of this rank run into its deadline (or into a peer's poison)?  (read after a stream synchronisation)
    bool exchange_failed() { if (ll_err && *ll_err) { fail(timeout_text()); return true; } return false; }
    // (how far this rank had come says which exchange the job stalled in: the counts are the same on every rank of a healthy job)
    std::string timeout_text() const {
        std::string what;
        if (ll_err) {
            const int phase = ll_err[1], peer = ll_err[2], idx = ll_err[3];
            if (phase == 1) what = "; first missing word: rank " + std::to_string(peer) + "'s contribution to element " + std::to_string(idx) + " of this rank's slice (that rank's kernel never stored it)";
            else if (phase == 2) what = "; first missing word: owner rank " + std::to_string(peer) + "'s sum for element " + std::to_string(idx) + " (that rank's kernel never answered)";
            else if (phase == 3) what = "; a peer's poison word arrived first (rank " + std::to_string(peer) + " was the one waited for: it, or a rank it waited for, had given up)";
        }
        return "a device-driven exchange timed out waiting for a peer rank (rank " + std::to_string(rank) + " of " + std::to_string(nranks) + " had launched " +
               std::to_string(ll_seq[0]) + " vector and " + std::to_string(ll_seq[1]) + " scalar exchanges; deadline " + std::to_string((int)(ll_timeout_s * 1e3)) + " ms" +
               what + "; " + std::to_string(ranks_on_my_device) + " rank(s) on this device holding " + std::to_string(queues_on_my_device) + " hardware queues)";
    }

    // Closing rendezvous: the last all-reduce launched its reduce / gather kernel asynchronously after its last host barrier, so a
    // peer's kernel may still be reading this rank's buffers when this rank gets here.  Drain the device, then meet the peers
    // once more (short time-out, the error flag tolerated: a failed job must still be able to leave) before anything is unmapped.
    void finalize() {
        if (finalized || !ctl) return;
        finalized = true;
        (void)hipDeviceSynchronize();
        if (!ready) return;                                    // the rendezvous never completed: nobody maps this rank's buffer
        const double keep = timeout_s;
        const std::string keep_err = err;
        timeout_s = std::min(timeout_s, 10.0);
        (void)barrier();                                       // (false when a peer failed or is gone: leave anyway)
        err = keep_err; timeout_s = keep;
    }
    bool finalized = false, ready = false;

    ~P2PComm() {
        finalize();
        for (int r = 0; r < nranks; ++r) if (r != rank && peer[r]) (void)hipIpcCloseMemHandle(peer[r]);
        for (int r = 0; r < nranks; ++r) if (r != rank && ll_peer[r]) (void)hipIpcCloseMemHandle(ll_peer[r]);
        ll_peer[rank] = nullptr;
        if (llbuf) (void)hipFree(llbuf);
        if (xbuf) (void)hipFree(xbuf);
        if (ll_err) (void)hipHostFree(ll_err);
        if (ctl) munmap(ctl, sizeof(P2PCtl));
    }
};
