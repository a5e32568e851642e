// sanitize/p2p_tsan_harness.cpp -- ThreadSanitizer run of the host side of the peer-to-peer communicator (pcr_p2p.h): the
// POSIX shared-memory rendezvous (O_EXCL creation, magic word written last, age / rank-count check), the generation barrier,
// the shared error flag and the closing rendezvous -- the lock-free code between the ranks of a job.
//
// The ranks run as THREADS of this process (TSan follows one process), with HIP mocked out (mock_hip/).  Every rank maps the
// control block itself, as in production; so that TSan sees ONE address per shared word, mmap / munmap of the block are routed
// (by macro, in this harness only) through a table that hands the same mapping to every rank.  Scenarios:
//   1. N ranks rendezvous (rank 0 deliberately late, a stale block of a "crashed job" left under the name), run 200 scalar and
//      vector all-reduces on the host-synchronised path (four barriers each), finalize, destroy;
//   2. the same with one rank reporting an error mid-way: every other rank must leave its barrier with an error, none may hang;
//   3. a rank that never shows up: the others time out of the rendezvous;
//   4. 8 ranks (the target machine) and 16 (the control block's table size) through scenario 1; 17 ranks are refused;
//   5. the queue budget of ranks that share a device (NOTES.md round 6: past 24 hardware queues per GPU, kernels of different
//      processes stop running side by side): 4 ranks on one device holding 9 queues each -> every rank switches the
//      device-driven exchange off and says why; 4 x 4 queues -> it stays on; 8 ranks on 8 devices holding 9 each -> on;
//      two devices, one of them shared by three ranks over the budget -> off on EVERY rank (one decision per job).
// Exit code 0 and no "WARNING: ThreadSanitizer" on stderr = clean (tests/test_sanitizers.py).
#include <sys/mman.h>
#include <sys/stat.h>

#include <atomic>
#include <cstdio>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

static std::mutex g_map_mu;
struct Mapping { void* p; size_t len; int refs; };
static std::map<ino_t, Mapping> g_maps;
static void* shared_mmap(void* addr, size_t len, int prot, int flags, int fd, off_t off) {
    struct stat sb;
    if (fstat(fd, &sb) != 0) return MAP_FAILED;
    std::lock_guard<std::mutex> lk(g_map_mu);
    auto it = g_maps.find(sb.st_ino);
    if (it != g_maps.end() && it->second.len == len) { it->second.refs++; return it->second.p; }
    void* p = mmap(addr, len, prot, flags, fd, off);
    if (p != MAP_FAILED) g_maps[sb.st_ino] = Mapping{p, len, 1};
    return p;
}
static int shared_munmap(void* p, size_t len) {
    std::lock_guard<std::mutex> lk(g_map_mu);
    for (auto it = g_maps.begin(); it != g_maps.end(); ++it)
        if (it->second.p == p) {
            if (--it->second.refs > 0) return 0;
            g_maps.erase(it);
            break;
        }
    return munmap(p, len);
}
#define mmap shared_mmap
#define munmap shared_munmap
#include "../pcr_p2p.h"
#undef mmap
#undef munmap

static int run_job(const char* name, int N, int fail_rank, int absent_rank, double timeout_s) {
    std::vector<std::thread> th;
    std::atomic<int> ok{0}, failed{0};
    for (int r = 0; r < N; ++r) {
        if (r == absent_rank) continue;
        th.emplace_back([&, r]() {
            if (r == 0) std::this_thread::sleep_for(std::chrono::milliseconds(30));      // the others find no (or a stale) block first
            P2PComm c;
            c.timeout_s = timeout_s;
            if (!c.init(name, r, N, 1000, 4, 0)) { failed++; return; }
            std::vector<float> v(1000, (float)(r + 1));
            std::vector<double> s(8, (double)(r + 1));
            bool good = true;
            for (int it = 0; it < 200 && good; ++it) {
                if (r == fail_rank && it == 57) { c.abort_peers(); good = false; break; }
                good = c.allreduce<float>(v.data(), v.size(), nullptr) && c.allreduce<double>(s.data(), s.size(), nullptr, true);
            }
            if (good) ok++; else failed++;
        });
    }
    for (auto& t : th) t.join();
    fprintf(stderr, "[harness] %-28s ranks %d: %d ok, %d failed\n", name, N, ok.load(), failed.load());
    return ok.load() * 100 + failed.load();
}

// scenario 5: ranks as threads, each on the device `bus_of(r)` with `queues` hardware queues; returns how many ranks ended with the
// device-driven exchange ON (x 100) + how many carry the budget's note
static int run_budget(const char* name, int N, int queues, const char* (*bus_of)(int)) {
    std::vector<std::thread> th;
    std::atomic<int> on{0}, noted{0}, failed{0};
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r]() {
            g_mock_bus = bus_of(r);
            P2PComm c;
            c.timeout_s = 30.0;
            c.my_queues = queues;
            if (!c.init(name, r, N, 1000, 4, (size_t)1 << 20)) { failed++; return; }
            if (c.ll_max_bytes) on++;
            if (c.ll_off_by_budget && !c.note.empty() && c.note.find("hardware queues") != std::string::npos) noted++;
        });
    for (auto& t : th) t.join();
    fprintf(stderr, "[harness] %-28s ranks %d x %d queues: device-driven exchange on for %d, switched off by the budget on %d, %d failed\n", name, N, queues,
            on.load(), noted.load(), failed.load());
    return failed.load() ? -1 : on.load() * 100 + noted.load();
}
static const char* one_device(int) { return "0000:05:00.0"; }
static const char* own_device(int r) {
    static const char* b[16] = {"0000:05:00.0", "0000:15:00.0", "0000:25:00.0", "0000:35:00.0", "0000:45:00.0", "0000:55:00.0", "0000:65:00.0", "0000:75:00.0",
                                "0000:85:00.0", "0000:95:00.0", "0000:a5:00.0", "0000:b5:00.0", "0000:c5:00.0", "0000:d5:00.0", "0000:e5:00.0", "0000:f5:00.0"};
    return b[r & 15];
}
static const char* three_and_one(int r) { return r < 3 ? "0000:05:00.0" : "0000:15:00.0"; }

int main() {
    int bad = 0;
    if (run_budget("/pcr_tsan_b1", 4, 9, one_device) != 4) bad |= 64;          // 36 queues on one GPU: off everywhere, every rank knows why
    if (run_budget("/pcr_tsan_b2", 4, 4, one_device) != 400) bad |= 128;       // 16: within the budget less its reserve
    if (run_budget("/pcr_tsan_b3", 8, 9, own_device) != 800) bad |= 256;       // one rank per GPU: nothing is shared, nothing is summed
    if (run_budget("/pcr_tsan_b4", 4, 9, three_and_one) != 4) bad |= 512;      // 27 on the shared device: the whole job leaves the device-driven path
    {   // a dead job's block under the name: magic set, old creation time, an error flag, posted words
        const char* name = "/pcr_tsan_h1";
        shm_unlink(name);
        int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, sizeof(P2PCtl)) == 0) {
            void* m = ::mmap(nullptr, sizeof(P2PCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                P2PCtl* c = static_cast<P2PCtl*>(m);
                c->created_ns = 1; c->nranks_expected = 4; c->error.store(1); c->posted[1].store(1);
                c->magic.store(PCR_P2P_MAGIC);
                ::munmap(m, sizeof(P2PCtl));
            }
        }
        if (fd >= 0) close(fd);
        if (run_job(name, 4, -1, -1, 20.0) != 400) bad |= 1;
    }
    if (run_job("/pcr_tsan_h2", 3, 1, -1, 20.0) != 3) bad |= 2;            // rank 1 fails: all three leave with an error
    if (run_job("/pcr_tsan_h3", 3, -1, 2, 1.0) != 2) bad |= 4;             // rank 2 never comes: the others time out
    shm_unlink("/pcr_tsan_h3");
    // the target machine's rank count, and the table's limit (PCR_P2P_MAXR): posted[] / handle[] / the barrier counter at 8 and 16
    if (run_job("/pcr_tsan_h4", 8, -1, -1, 60.0) != 800) bad |= 8;
    if (run_job("/pcr_tsan_h5", PCR_P2P_MAXR, -1, -1, 60.0) != 100 * PCR_P2P_MAXR) bad |= 16;
    {   // one more rank than the table holds: refused by every rank, nobody waits
        std::atomic<int> refused{0};
        std::vector<std::thread> th;
        for (int r = 0; r < 2; ++r) th.emplace_back([&, r]() { P2PComm c; c.timeout_s = 2.0; if (!c.init("/pcr_tsan_h6", r, PCR_P2P_MAXR + 1, 10, 4, 0)) refused++; });
        for (auto& t : th) t.join();
        if (refused.load() != 2) bad |= 32;
    }
    fprintf(stderr, bad ? "[harness] FAILED (%d)\n" : "[harness] all scenarios behaved\n", bad);
    return bad ? 1 : 0;
}
