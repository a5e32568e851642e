// omp-pmf-train -- drop-in replacement of the reference's training CLI (pmf-train.cpp) on top of
// the C ABI of libprimalcr (MI355X).  Same flags, defaults, positional rules, default model name,
// log lines, model file and U.txt / V.txt side files (pmf-train.cpp:8-27, 29-135, 247-314).
//
// Extensions use long options the reference would reject anyway (unknown option -> usage, exit 1):
//   --f64          fp64 storage on the device (default: fp32 storage, fp64 accumulation)
//   --device N     HIP device ordinal
//   --init-model F warm start: take the initial U, V from a model file instead of initial()
//                  (the reference only has a commented-out text-file variant, pmf-train.cpp:262-263)
//   --cache F      binary side-car of the parsed data set: read F if it matches the text files' size and
//                  modification time, else parse the text and (best effort) write F
//   --snapshot-every N  also write <model>.iter<k> after every N-th outer iteration (with --gpus N too: every rank deposits its rows)
//   --cg-iters N / --cg-tol X  truncated-Newton knobs (the reference hard-codes 10 / 0.01); --cg-iters k with a
//                       small --cg-tol makes the U step an exact Newton step
//   --gpus N       user-shard the training over N GPUs of this node: the data set is parsed once, then one
//                  worker process per GPU is forked BEFORE anything touches a GPU; worker q owns the users
//                  pcr_partition_users gives rank q, V and the CG vectors are replicated, the V-gradient and every
//                  Hessian-vector product are all-reduced.  Worker 0 logs; at the end every worker deposits its rows of U
//                  (the "all-gather of U shards") and the parent writes the model and U.txt / V.txt.
//   --devices a,b,..  the HIP device of every rank (default 0..N-1)
//   --comm rccl|p2p   exchange step: ncclAllReduce over RCCL (default) or the direct peer-to-peer reduce-scatter /
//                     all-gather of pcr_solver_comm_init_p2p
//   --tune key=value  a launch knob of pcr_tune() (repeatable)
//   --timing          one "[timing] load_s=... init_s=... create_s=... train_s=... iter_s=... eval_s=... write_s=... wall_s=..." line
//                     on stderr at the end: where the run's wall time went (iter_s = the reference's own "Iter k time" clock,
//                     pcrpp.cpp:874-881; eval_s = train_s - iter_s; wall_s = from the first line of main to the last)
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <csignal>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "primalcr.h"

static void exit_with_help() {
    printf(
        "Usage: omp-pmf-train [options] data_dir [model_filename]\n"
        "options:\n"
        "    -s type : set type of solver (default 2)\n"
        "    	 1 -- PirmalCR\n"
        "    	 2 -- PrimalCR++\n"
        "    -k rank : set the rank (default 10)\n"
        "    -n threads : set the number of threads (default 4)\n"
        "    -l lambda : set the regularization parameter lambda (default 5000)\n"
        "    -t max_iter: set the number of iterations (default 10)\n"
        "    -p do_predict: compute training/testing error & NDCG at each iteration or not (default 1)\n"
        "    --f64 : keep U, V in fp64 on the GPU (default fp32 storage, fp64 accumulation)\n"
        "    --device id : GPU to use (default 0)\n"
        "    --init-model file : warm start from a model file\n"
        "    --cache file : binary cache of the parsed data set (rebuilt when the text files change)\n"
        "    --snapshot-every n : also write <model>.iter<k> after every n-th iteration\n"
        "    --cg-iters n : CG iterations per Newton step at most (default 10, the reference's constant)\n"
        "    --cg-tol x : CG residual tolerance relative to ||g|| (default 0.01, the reference's constant)\n"
        "    --gpus n : shard the users over n GPUs of this node, one worker process per GPU (default 1)\n"
        "    --devices a,b,.. : HIP device of every rank (default 0..n-1)\n"
        "    --comm rccl|p2p : all-reduce through RCCL (default) or direct peer-to-peer buffers\n"
        "    --tune key=value : launch knob (pcr_tune)\n"
        "    --timing : print the phases of the run's wall time on stderr\n");
    exit(1);
}

// --snapshot-every: the training loop hands every log line to this callback (same text as the default stdout logger);
// after every n-th "Iter k ..." line the current factors are written to <model>.iter<k> in the model-file format.
struct SnapCtx {
    pcr_solver* s; int every; std::string model; int64_t d1, d2; int k;
    std::vector<double>*U, *V; bool failed;
};
static void snap_log(void* vctx, const char* line) {
    SnapCtx* c = static_cast<SnapCtx*>(vctx);
    fputs(line, stdout); fputc('\n', stdout); fflush(stdout);
    int it = 0;
    if (c->every <= 0 || sscanf(line, "Iter %d time", &it) != 1 || it <= 0 || it % c->every != 0) return;
    const std::string path = c->model + ".iter" + std::to_string(it);
    if (pcr_solver_get_factors(c->s, c->U->data(), c->V->data()) != PCR_OK ||
        pcr_model_save(path.c_str(), c->U->data(), c->d1, c->V->data(), c->d2, c->k) != PCR_OK) {
        fprintf(stderr, "snapshot %s: %s\n", path.c_str(), pcr_last_error());
        c->failed = true;
    }
}

// the thread that brings the HIP runtime up while main() parses the input (one-GPU runs): joined before any way out of the process
static std::thread g_warm;
static void join_warm() { if (g_warm.joinable()) g_warm.join(); }
static bool g_lanes_given = false;                                        // --tune lanes=... on the command line

static void die_with(const char* what, const std::string& msg) {
    join_warm();
    fprintf(stderr, "%s: %s\n", what, msg.c_str());
    exit(1);
}
static void die(const char* what) { die_with(what, pcr_last_error()); }

// side files (pmf-train.cpp:208-227, 276-295) and the model (pmf-train.cpp:297-310)
static void write_outputs(const pcr_params& param, const std::string& model, const std::vector<double>& U, const std::vector<double>& V,
                          int64_t d1, int64_t d2) {
    const int k = param.k;
    std::string suffix = param.solver_type == PCR_SOLVER_PCR ? std::to_string(static_cast<int>(param.lambda)) : "";
    // Same bytes as the reference's `f << M[a][b]` (an ofstream at its default precision = printf's "%g" = std::to_chars in its
    // general format at precision 6, which the standard defines through that printf conversion), formatted by up to 16 threads
    // into per-thread buffers that go to the file in row order: 48 M numbers (the Netflix shape's U) are seconds, not a minute.
    auto dump = [&](const char* name, const std::vector<double>& M, int64_t rows) {
        std::cout << name << " matrix of size " << rows << ", " << k << std::endl;
        const std::string path = std::string(name) + suffix + ".txt";
        FILE* f = fopen(path.c_str(), "wb");
        // (a side file that cannot be written is an error, not a silent omission: a full disk must not end in exit code 0)
        if (!f) die_with("output", "can't open " + path + ": " + strerror(errno));
        bool ok = true;
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), rows * k / 65536 + 1));
        const int64_t rows_per_round = std::max<int64_t>(1, ((int64_t)1 << 20) / std::max(1, k));      // ~1 M numbers (16 MB of text at most) per thread and round
        for (int64_t r0 = 0; r0 < rows; r0 += rows_per_round * T) {
            std::vector<std::string> out((size_t)T);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    const int64_t a0 = std::min(rows, r0 + rows_per_round * t), a1 = std::min(rows, a0 + rows_per_round);
                    std::string& o = out[(size_t)t];
                    o.resize((size_t)(a1 - a0) * (size_t)k * 16 + 16);
                    char* p = &o[0];
                    for (int64_t a = a0; a < a1; ++a)
                        for (int b = 0; b < k; ++b) { p = std::to_chars(p, p + 16, M[a * k + b], std::chars_format::general, 6).ptr; *p++ = b < k - 1 ? ' ' : '\n'; }
                    o.resize((size_t)(p - &o[0]));
                });
            for (auto& x : th) x.join();
            for (int t = 0; t < T; ++t)
                if (!out[(size_t)t].empty()) ok = fwrite(out[(size_t)t].data(), 1, out[(size_t)t].size(), f) == out[(size_t)t].size() && ok;
        }
        ok = fclose(f) == 0 && ok;
        if (!ok) die_with("output", "short write to " + path + ": " + strerror(errno));
    };
    dump("U", U, d1);
    dump("V", V, d2);
    if (pcr_model_save(model.c_str(), U.data(), d1, V.data(), d2, k) != PCR_OK) die("model");
}

// ---- --gpus N: one worker process per GPU -------------------------------------------------------------------------------
// Shared between the parent and its workers (anonymous MAP_SHARED, made before the fork): the RCCL id, the result factors.
struct SharedHdr {
    std::atomic<int> id_ready, failed, nranks_reported;
    std::atomic<int> done[16];               // rank q has deposited its rows of U (and rank 0 V): all of them before the model is written
    std::atomic<int> snap[16];               // --snapshot-every: the last outer iteration whose factors rank q has deposited
    unsigned char nccl_id[128];
    char shm_name[64];
};

// the parent's children, for the signal handler: killing the parent must not leave GPU workers behind
static volatile sig_atomic_t g_nkids = 0;
static pid_t g_kids[16];
static volatile sig_atomic_t g_signalled = 0;
static void forward_signal(int sig) {
    g_signalled = sig;
    for (int i = 0; i < g_nkids; ++i) if (g_kids[i] > 0) kill(g_kids[i], SIGTERM);
}

// --snapshot-every with --gpus N (pmf-train.cpp:297-310 writes the model only once, at the end: a long 8-GPU run has no
// resume point).  pcr_train hands every rank the log lines; at every n-th "Iter k" line each rank deposits its own rows of U
// (rank 0 also V) in the shared block and publishes k; rank 0 waits until every rank has published k and writes
// <model>.iter<k>.  No rank can overwrite its rows with a later iteration's meanwhile: the next outer iteration needs
// all-reduces that rank 0, busy here, has not joined yet.
struct MultiSnap {
    pcr_solver* s; int every, rank, nranks; struct SharedHdr* hdr; double* Ush; double* Vsh;
    int64_t d1, d2, first, nu; int k; std::string model; bool failed;
};
static void multi_snap_log(void* vctx, const char* line);

// body of worker `rank`: everything that touches a GPU happens here, after the fork
static int worker(const pcr_dataset* ds, pcr_params param, int rank, int nranks, const std::string& comm_kind, SharedHdr* hdr,
                  double* Ush, double* Vsh, const std::vector<double>& U0, const std::vector<double>& V0, int64_t d1, int64_t d2,
                  int snapshot_every, const std::string& model) {
    auto fail = [&](const char* what) { fprintf(stderr, "[rank %d] %s: %s\n", rank, what, pcr_last_error()); hdr->failed.store(1); return 1; };
    const auto t0 = std::chrono::steady_clock::now();
    pcr_solver* s = nullptr;
    if (pcr_solver_create(ds, &param, rank, nranks, &s) != PCR_OK) return fail("solver");
    if (comm_kind == "p2p") {
        if (pcr_solver_comm_init_p2p(s, hdr->shm_name) != PCR_OK) return fail("comm (p2p)");
    } else {
        if (rank == 0) {
            if (pcr_comm_unique_id(hdr->nccl_id) != PCR_OK) return fail("comm id");
            hdr->id_ready.store(1, std::memory_order_release);
        }
        for (int spins = 0; !hdr->id_ready.load(std::memory_order_acquire); ++spins) {
            if (hdr->failed.load() || spins > 120 * 1000) { fprintf(stderr, "[rank %d] no communicator id from rank 0\n", rank); return 1; }
            usleep(1000);
        }
        if (pcr_solver_comm_init(s, hdr->nccl_id) != PCR_OK) return fail("comm (rccl)");
    }
    if (rank == 0) hdr->nranks_reported.store(pcr_solver_comm_nranks(s));
    {
        int64_t f0 = 0, n0 = 0, z0 = 0;
        pcr_solver_shard(s, &f0, &n0, &z0);
        fprintf(stderr, "[rank %d] device %d: users [%ld, %ld), %ld ratings\n", rank, param.device, (long)f0, (long)(f0 + n0), (long)z0);
    }
    if (pcr_solver_set_factors(s, U0.data(), V0.data()) != PCR_OK) return fail("set_factors");
    int64_t first = 0, nu = 0;
    pcr_solver_shard(s, &first, &nu, nullptr);
    MultiSnap snap{s, snapshot_every, rank, nranks, hdr, Ush, Vsh, d1, d2, first, nu, param.k, model, false};
    if (pcr_train(s, snapshot_every > 0 ? multi_snap_log : nullptr, &snap, nullptr) != PCR_OK) return fail("train");
    if (snap.failed) { hdr->failed.store(1); return 1; }
    // "all-gather of U shards" (SURVEY 8e): every rank deposits its own rows; V is replicated, rank 0 deposits it
    std::vector<double> Uf((size_t)d1 * param.k), Vf((size_t)d2 * param.k);
    if (pcr_solver_get_factors(s, Uf.data(), rank == 0 ? Vf.data() : nullptr) != PCR_OK) return fail("get_factors");
    memcpy(Ush + first * param.k, Uf.data() + first * param.k, (size_t)nu * param.k * sizeof(double));
    if (rank == 0) {
        memcpy(Vsh, Vf.data(), Vf.size() * sizeof(double));
        printf("Wall-time: %lg secs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        fflush(stdout);
    }
    hdr->done[rank].store(1, std::memory_order_release);
    pcr_solver_destroy(s);
    return 0;
}

static void multi_snap_log(void* vctx, const char* line) {
    MultiSnap* c = static_cast<MultiSnap*>(vctx);
    if (c->rank == 0) { fputs(line, stdout); fputc('\n', stdout); fflush(stdout); }
    int it = 0;
    if (c->failed || sscanf(line, "Iter %d time", &it) != 1 || it <= 0 || it % c->every != 0) return;
    if (pcr_solver_get_factors_local(c->s, c->Ush + c->first * c->k, c->rank == 0 ? c->Vsh : nullptr) != PCR_OK) {
        fprintf(stderr, "[rank %d] snapshot at iteration %d: %s\n", c->rank, it, pcr_last_error());
        c->failed = true; c->hdr->failed.store(1);
        return;
    }
    c->hdr->snap[c->rank].store(it, std::memory_order_release);
    if (c->rank != 0) return;
    for (int q = 0; q < c->nranks; ++q)
        for (long spins = 0; c->hdr->snap[q].load(std::memory_order_acquire) < it; ++spins) {
            if (c->hdr->failed.load() || spins > 120L * 1000) {
                fprintf(stderr, "[rank 0] snapshot at iteration %d: rank %d did not deposit its rows\n", it, q);
                c->failed = true; c->hdr->failed.store(1);
                return;
            }
            usleep(1000);
        }
    const std::string path = c->model + ".iter" + std::to_string(it);
    if (pcr_model_save(path.c_str(), c->Ush, c->d1, c->Vsh, c->d2, c->k) != PCR_OK) {
        fprintf(stderr, "snapshot %s: %s\n", path.c_str(), pcr_last_error());
        c->failed = true; c->hdr->failed.store(1);
    }
}

static int train_multi(const pcr_dataset* ds, const pcr_params& param, int gpus, const std::vector<int>& devices,
                       const std::string& comm_kind, std::vector<double>& U, std::vector<double>& V, int64_t d1, int64_t d2,
                       int snapshot_every, const std::string& model) {
    const size_t nU = (size_t)d1 * param.k, nV = (size_t)d2 * param.k;
    const size_t bytes = sizeof(SharedHdr) + (nU + nV) * sizeof(double) + 64;
    void* mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (mem == MAP_FAILED) { perror("mmap"); return 1; }
    SharedHdr* hdr = new (mem) SharedHdr();
    hdr->id_ready.store(0); hdr->failed.store(0); hdr->nranks_reported.store(0);
    for (auto& d : hdr->done) d.store(0);
    for (auto& d : hdr->snap) d.store(0);
    {   // the control block's name: pid + something a bystander cannot predict
        unsigned long long rnd = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        if (FILE* ur = fopen("/dev/urandom", "rb")) { if (fread(&rnd, sizeof rnd, 1, ur) != 1) rnd ^= (unsigned long long)getpid() << 32; fclose(ur); }
        snprintf(hdr->shm_name, sizeof hdr->shm_name, "/pcr_p2p_%d_%016llx", (int)getpid(), rnd);
    }
    double* Ush = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(hdr + 1) + 63) & ~(uintptr_t)63);
    double* Vsh = Ush + nU;
    fflush(stdout); fflush(stderr);
    bool shared_device = false;
    for (size_t a = 0; a < devices.size(); ++a)
        for (size_t b = a + 1; b < devices.size(); ++b) shared_device = shared_device || devices[a] == devices[b];
    const bool pcr_tune_is_default_lanes = !g_lanes_given;
    if (shared_device && pcr_tune_is_default_lanes) fprintf(stderr, "omp-pmf-train: several ranks share a device: one stream per rank (--tune lanes=1)\n");
    std::vector<pid_t> kids;
    struct sigaction sa_new, sa_int, sa_term;
    memset(&sa_new, 0, sizeof sa_new);
    sa_new.sa_handler = forward_signal;                   // (no SA_RESTART: wait() returns EINTR and the loop below sees g_signalled)
    sigemptyset(&sa_new.sa_mask);
    sigaction(SIGINT, &sa_new, &sa_int);
    sigaction(SIGTERM, &sa_new, &sa_term);
    for (int q = 0; q < gpus; ++q) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); for (pid_t k : kids) kill(k, SIGKILL); return 1; }
        if (pid == 0) {
            signal(SIGINT, SIG_DFL); signal(SIGTERM, SIG_DFL);
            pcr_params p = param;
            p.device = devices.empty() ? q : devices[q];
            // Ranks that share a device (a rehearsal of the N-rank path on fewer GPUs) share its hardware queues too: every rank
            // keeps to its solver's stream instead of probing and using side lanes of its own (pcr_tune "lanes" = 1, unless
            // the command line says otherwise) -- five processes that each claim half a dozen queues, with spinning probe kernels
            // measuring each other, is what the one-rank-per-GPU layout never sees.
            if (shared_device && pcr_tune_is_default_lanes) (void)pcr_tune("lanes", "1");
            const int rc = worker(ds, p, q, gpus, comm_kind, hdr, Ush, Vsh, U, V, d1, d2, snapshot_every, model);
            fflush(stdout); fflush(stderr);
            _exit(rc);
        }
        kids.push_back(pid);
        g_kids[g_nkids] = pid; g_nkids = g_nkids + 1;
    }
    // the parent never touches a GPU: it waits; the first worker that fails takes the others with it (a rank blocked in an
    // RCCL collective whose peer died would wait forever)
    int bad = 0;
    for (size_t left = kids.size(); left > 0;) {
        int st = 0;
        const pid_t pid = wait(&st);
        if (pid < 0) {
            if (errno == EINTR) {                          // a signal for the parent: its handler has told the workers; keep reaping
                if (g_signalled && !bad) { bad = 1; hdr->failed.store(1); }
                continue;
            }
            perror("wait");                                // ECHILD etc.: the job's outcome is unknown -> failure
            bad = 1;
            hdr->failed.store(1);
            for (pid_t k : kids) kill(k, SIGTERM);
            break;
        }
        --left;
        const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
        if (!ok && !bad) {
            bad = 1;
            hdr->failed.store(1);
            for (pid_t k : kids) if (k != pid) kill(k, SIGTERM);
        }
    }
    g_nkids = 0;
    sigaction(SIGINT, &sa_int, nullptr);
    sigaction(SIGTERM, &sa_term, nullptr);
    shm_unlink(hdr->shm_name);                           // (rank 0 unlinks it once everyone is attached; a job that failed earlier did not)
    if (g_signalled) bad = 1;
    for (int q = 0; q < gpus && !bad; ++q)
        if (!hdr->done[q].load(std::memory_order_acquire)) { fprintf(stderr, "omp-pmf-train: rank %d left without depositing its factors\n", q); bad = 1; }
    if (bad) { fprintf(stderr, "omp-pmf-train: a GPU worker failed\n"); munmap(mem, bytes); return 1; }
    if (hdr->nranks_reported.load() != gpus) {
        fprintf(stderr, "omp-pmf-train: the communicator reports %d ranks, expected %d\n", hdr->nranks_reported.load(), gpus);
        munmap(mem, bytes);
        return 1;
    }
    memcpy(U.data(), Ush, nU * sizeof(double));
    memcpy(V.data(), Vsh, nV * sizeof(double));
    munmap(mem, bytes);
    return 0;
}

int main(int argc, char** argv) {
    const auto t_main = std::chrono::steady_clock::now();
    auto t_lap = t_main;
    auto lap = [&]() { const auto n = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double>(n - t_lap).count(); t_lap = n; return d; };
    bool timing = false;
    double load_s = 0, init_s = 0, create_s = 0, train_s = 0, iter_s = 0, write_s = 0;
    pcr_params param;
    pcr_params_default(&param);
    std::string init_model, cache, comm_kind = "rccl";
    std::vector<int> devices;
    int snapshot_every = 0, gpus = 1;
    int i;
    for (i = 1; i < argc; i++) {                       // pmf-train.cpp:36-108
        if (argv[i][0] != '-') break;
        if (!strcmp(argv[i], "--f64")) { param.precision = PCR_F64; continue; }
        if (!strcmp(argv[i], "--timing")) { timing = true; continue; }
        if (++i >= argc) exit_with_help();
        if (!strcmp(argv[i - 1], "--device")) { param.device = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--init-model")) { init_model = argv[i]; continue; }
        if (!strcmp(argv[i - 1], "--cache")) { cache = argv[i]; continue; }
        if (!strcmp(argv[i - 1], "--snapshot-every")) { snapshot_every = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--cg-iters")) { param.cg_max_iter = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--cg-tol")) { param.cg_tol = atof(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--gpus")) { gpus = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--comm")) { comm_kind = argv[i]; continue; }
        if (!strcmp(argv[i - 1], "--devices")) {
            for (const char* q = argv[i]; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
            continue;
        }
        if (!strcmp(argv[i - 1], "--tune")) {
            std::string kv = argv[i];
            const size_t eq = kv.find('=');
            if (kv.compare(0, 6, "lanes=") == 0) g_lanes_given = true;
            if (eq == std::string::npos || pcr_tune(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str()) != PCR_OK) {
                fprintf(stderr, "--tune %s: %s\n", argv[i], eq == std::string::npos ? "expected key=value" : pcr_last_error());
                return 1;
            }
            continue;
        }
        switch (argv[i - 1][1]) {
            case 's': param.solver_type = atoi(argv[i]); break;
            case 'k': param.k = atoi(argv[i]); break;
            case 'n': param.threads = atoi(argv[i]); break;
            case 'l': param.lambda = atof(argv[i]); break;
            case 't': param.maxiter = atoi(argv[i]); break;
            case 'p': param.do_predict = atoi(argv[i]); break;
            case 'q': param.verbose = atoi(argv[i]); break;
            // parsed-but-unused by the PCR/PCR++ path in the reference (pmf-train.cpp:59-102)
            case 'r': case 'T': case 'e': case 'B': case 'm': case 'u': case 'd': case 'N': break;
            default:
                fprintf(stderr, "unknown option: -%c\n", argv[i - 1][1]);
                exit_with_help();
        }
    }
    if (param.do_predict != 0) param.verbose = 1;
    if (i >= argc) exit_with_help();
    std::string input = argv[i], model;
    if (i < argc - 1) model = argv[i + 1];
    else {                                             // pmf-train.cpp:120-133
        std::string d = input;
        while (!d.empty() && d.back() == '/') d.pop_back();
        size_t p = d.rfind('/');
        model = (p == std::string::npos ? d : d.substr(p + 1)) + ".model";
    }
    if (param.solver_type != PCR_SOLVER_PCR && param.solver_type != PCR_SOLVER_PCRPP) {
        fprintf(stderr, "Error: wrong solver type (%d)!\n", param.solver_type);   // pmf-train.cpp:331-333
        return 0;
    }
    // the reference opens the model file BEFORE training (pmf-train.cpp:252-259)
    FILE* fp = fopen(model.c_str(), "wb");
    if (!fp) { fprintf(stderr, "can't open output file %s\n", model.c_str()); return 1; }
    fclose(fp);

    // One GPU: the HIP runtime and the library's code object come up on a second thread while this one parses the ratings and draws
    // the initial factors (with --gpus N nothing may touch a GPU before the workers are forked).  Its result does not matter here:
    // pcr_solver_create reports a missing device itself.
    std::thread& warm = g_warm;
    if (gpus == 1) warm = std::thread([dev = devices.empty() ? param.device : devices[0]]() { (void)pcr_device_warmup(dev); });
    struct Joiner { ~Joiner() { join_warm(); } } join_at_return;
    pcr_dataset* ds = nullptr;
    (void)lap();
    if ((cache.empty() ? pcr_dataset_load_mt(input.c_str(), param.threads, &ds)
                       : pcr_dataset_load_cached(input.c_str(), param.threads, cache.c_str(), &ds)) != PCR_OK) die("load");
    load_s = lap();
    // (every way out of main from here on releases the data set and the solver: found by LeakSanitizer on the failure paths)
    struct Holder { pcr_dataset* ds; pcr_solver* s; ~Holder() { if (s) pcr_solver_destroy(s); if (ds) pcr_dataset_free(ds); } } hold{ds, nullptr};
    int64_t d1, d2, nnz, tnnz;
    pcr_dataset_dims(ds, &d1, &d2, &nnz, &tnnz);
    const int k = param.k;
    std::vector<double> U((size_t)d1 * k), V((size_t)d2 * k);
    pcr_initial(U.data(), d1, k);                      // pmf-train.cpp:264-266
    pcr_initial(V.data(), d2, k);
    if (!init_model.empty()) {                         // warm start
        int64_t m1, m2, kk;
        if (pcr_model_load(init_model.c_str(), &m1, &m2, &kk, nullptr, nullptr) != PCR_OK) die("init-model");
        if (m1 != d1 || m2 != d2 || kk != k) {
            fprintf(stderr, "init-model %s is %ld x %ld / %ld x %ld, expected %ld x %d / %ld x %d\n", init_model.c_str(),
                    (long)m1, (long)kk, (long)m2, (long)kk, (long)d1, k, (long)d2, k);
            return 1;
        }
        if (pcr_model_load(init_model.c_str(), &m1, &m2, &kk, U.data(), V.data()) != PCR_OK) die("init-model");
    }
    init_s = lap();
    std::cout << "the rank is " << k << std::endl;
    std::cout << "the number of rows is " << d1 << " and the number of cols is " << d2 << std::endl;
    if (param.solver_type == PCR_SOLVER_PCRPP) { std::cout << nnz << std::endl; std::cout << "starts!" << std::endl; }
    else std::cout << "nnz: " << nnz << std::endl;

    if (gpus < 1 || gpus > 16 || (comm_kind != "rccl" && comm_kind != "p2p") || (!devices.empty() && (int)devices.size() != gpus)) {
        fprintf(stderr, "--gpus must be 1..16, --comm rccl or p2p, --devices one ordinal per rank\n");
        return 1;
    }
    auto report = [&]() {
        if (!timing) return;
        fprintf(stderr, "[timing] load_s=%.4f init_s=%.4f create_s=%.4f train_s=%.4f iter_s=%.4f eval_s=%.4f write_s=%.4f wall_s=%.4f\n", load_s, init_s,
                create_s, train_s, iter_s, std::max(0.0, train_s - iter_s), write_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count());
    };
    if (gpus > 1) {
        (void)lap();
        const int rc = train_multi(ds, param, gpus, devices, comm_kind, U, V, d1, d2, snapshot_every, model);
        if (rc != 0) return rc;
        train_s = lap();                                  // (workers: solver creation + training + deposit; rank 0 prints its own split)
        write_outputs(param, model, U, V, d1, d2);
        write_s = lap();
        report();
        return 0;
    }
    auto t0 = std::chrono::steady_clock::now();
    (void)lap();
    if (warm.joinable()) warm.join();
    const double warm_wait_s = lap();                     // what was left of HIP runtime initialisation + code-object load after load / init
    pcr_solver* s = nullptr;
    if (!devices.empty()) param.device = devices[0];
    if (pcr_solver_create(ds, &param, 0, 1, &s) != PCR_OK) { fprintf(stderr, "solver: %s\n", pcr_last_error()); return 1; }
    hold.s = s;
    auto fail = [](const char* what) { fprintf(stderr, "%s: %s\n", what, pcr_last_error()); return 1; };
    const double solver_s = lap();                        // pcr_solver_create alone (its phases: pcr_solver_setup_phase)
    if (pcr_solver_set_factors(s, U.data(), V.data()) != PCR_OK) return fail("set_factors");
    create_s = warm_wait_s + solver_s + lap();
    if (timing) {
        // where solver creation went: the join with the thread that brought the HIP runtime up, then the library's own phases
        fprintf(stderr, "[timing-create] wait_for_runtime_s=%.4f solver_create_s=%.4f set_factors_s=%.4f", warm_wait_s, solver_s, create_s - solver_s - warm_wait_s);
        const char* name = nullptr; double ms = 0.0;
        for (int ph = 0; pcr_solver_setup_phase(s, ph, &name, &ms) == PCR_OK; ++ph) {
            std::string key = name;
            for (char& ch : key) if (ch == ' ' || ch == ',' || ch == '-') ch = '_';
            fprintf(stderr, " %s=%.4f", key.c_str(), ms / 1e3);
        }
        fprintf(stderr, "\n");
    }
    SnapCtx snap{s, snapshot_every, model, d1, d2, k, &U, &V, false};
    std::vector<pcr_iter_stats> hist((size_t)std::max(0, param.maxiter) + 1);
    if (pcr_train(s, snapshot_every > 0 ? snap_log : nullptr, &snap, hist.data()) != PCR_OK) return fail("train");
    if (snap.failed) return 1;
    train_s = lap();
    iter_s = hist.back().seconds;                         // cumulative, the clock scope of pcrpp.cpp:874-881
    if (pcr_solver_get_factors(s, U.data(), V.data()) != PCR_OK) return fail("get_factors");
    printf("Wall-time: %lg secs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    write_outputs(param, model, U, V, d1, d2);
    write_s = lap();
    report();
#ifndef PCR_CLI_PLAIN_EXIT
    // Everything this run produces is on disk and flushed: leave without tearing the solver and the HIP runtime down piece by piece
    // (streams, queues, code objects: ~0.1 s that only a process about to exit pays; the driver reclaims them with the process).
    // Invariant: _exit skips `hold` and `join_at_return` -- the warm-up thread was joined before pcr_solver_create above, and every
    // output has been written AND checked (write_outputs leaves through die_with on a failed open / write / close).
    fflush(stdout); fflush(stderr);
    _exit(0);
#endif
    return 0;
}
