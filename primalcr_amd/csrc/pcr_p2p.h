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
// Why host barriers: xGMI peers see each other's plain stores reliably at kernel boundaries; the vectors are 1.6-109 MB, so
// two ~10 us host round trips per all-reduce are noise on the shapes where N > 1 pays (Netflix-/Yahoo-shaped), and a rank
// that dies or times out raises the shared error flag, which every peer polls in its barrier: no rank waits forever on a
// lost peer (with RCCL the surviving ranks block inside the collective).
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

#define PCR_P2P_MAGIC 0x50435250325033ull     // "PCRP2P3"
struct P2PCtl {                               // in POSIX shared memory, created (O_EXCL) and zero-filled by rank 0
    std::atomic<uint64_t> magic;              // written LAST by rank 0: the block is ready
    uint64_t created_ns;                      // CLOCK_REALTIME at creation: a block older than the rendezvous time-out is a dead job's
    uint32_t nranks_expected;
    std::atomic<uint32_t> count, gen, attached;
    std::atomic<int32_t> error;
    std::atomic<uint32_t> posted[PCR_P2P_MAXR];
    hipIpcMemHandle_t handle[PCR_P2P_MAXR];
    uint64_t bytes[PCR_P2P_MAXR];
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

struct P2PComm {
    int rank = 0, nranks = 1;
    P2PCtl* ctl = nullptr;
    std::string shm_name;
    char* xbuf = nullptr;                     // this rank's exchange buffer: X[2][cap] | Y[2][slice_cap] | S[2][64 doubles]
    char* peer[PCR_P2P_MAXR] = {};
    size_t cap_bytes = 0, slice_bytes = 0;    // one X / one Y (bytes)
    uint64_t seq = 0;
    double timeout_s = 120.0;
    std::string err;

    static constexpr size_t SCAL_BYTES = 64 * sizeof(double);
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
    bool init(const char* name, int rank_, int nranks_, size_t elems_max, size_t elt) {
        rank = rank_; nranks = nranks_;
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
                        const uint64_t age = now_ns() - c->created_ns;
                        if (c->magic.load(std::memory_order_acquire) == PCR_P2P_MAGIC && c->nranks_expected == (uint32_t)nranks &&
                            age < (uint64_t)(timeout_s * 1e9)) { ctl = c; break; }
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
        if (hipMalloc((void**)&xbuf, total_bytes()) != hipSuccess) return fail("hipMalloc of the exchange buffer failed");
        if (hipMemset(xbuf, 0, total_bytes()) != hipSuccess) return fail("hipMemset of the exchange buffer failed");
        // (dmabuf IPC: on hosts whose driver has no legacy IPC mode the process must run with HSA_ENABLE_IPC_MODE_LEGACY=0 --
        // an environment prerequisite of the ROCm runtime, listed in include/primalcr.h; the library itself reads no variable)
        if (hipIpcGetMemHandle(&ctl->handle[rank], xbuf) != hipSuccess) return fail("hipIpcGetMemHandle failed (on dmabuf-only hosts run with HSA_ENABLE_IPC_MODE_LEGACY=0)");
        ctl->bytes[rank] = total_bytes();
        ctl->posted[rank].store(1, std::memory_order_release);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < nranks; ++r)
            while (!ctl->posted[r].load(std::memory_order_acquire)) {
                if (ctl->error.load()) return fail("a peer rank reported an error during the rendezvous");
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("rendezvous timed out");
                std::this_thread::yield();
            }
        for (int r = 0; r < nranks; ++r) {
            if (r == rank) { peer[r] = xbuf; continue; }
            if (ctl->bytes[r] != total_bytes()) return fail("ranks disagree on the exchange buffer size");
            void* p = nullptr;
            if (hipIpcOpenMemHandle(&p, ctl->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
                return fail("hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s buffer failed");
            peer[r] = static_cast<char*>(p);
        }
        ctl->attached.fetch_add(1);
        if (!barrier()) return false;
        if (rank == 0) shm_unlink(name);      // everyone is attached: the name can go (the mapping lives on)
        ready = true;
        return true;
    }

    // in-place sum of buf[0, n) over the ranks; stream-ordered on st from the caller's point of view
    template <typename X>
    bool allreduce(X* buf, size_t n, hipStream_t st, bool scalars = false) {
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
        if (xbuf) (void)hipFree(xbuf);
        if (ctl) munmap(ctl, sizeof(P2PCtl));
    }
};
