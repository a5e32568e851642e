/*
 * primalcr.h -- C ABI of libprimalcr.so: the MI355X-native PrimalCR / PrimalCR++
 * training path (hand-written HIP for gfx950) behind the reference's solver API.
 *
 * Drop-in boundary.  The reference's solver entry points are
 *     extern "C" void pcrpp(smat_t&, mat_t&, mat_t&, testset_t&, parameter&)  pmf.h:52-56, pcrpp.cpp:841
 *     extern "C" void pcr  (smat_t&, mat_t&, mat_t&, testset_t&, parameter&)  pmf.h:54,    pcr.cpp:616
 * whose arguments are C++ references to STL classes -- not a real C ABI.  This
 * header is the genuine C ABI for the same step: plain pointers and sizes, caller
 * owned factor buffers (fp64, row-major, exactly the reference's mat_t payload),
 * integer status codes (the reference returns void and never reports errors).
 * INTEGRATION.md shows the few lines a maintainer adds to pmf-train.cpp to call it.
 *
 * Every entry point cites the reference interface it replaces (file:line relative
 * to the reference checkout).
 *
 * Conventions
 *   - all matrices row-major, 0-based; U is d1 x k (users), V is d2 x k (items)
 *   - ratings: user-major CSR in the reference's SparseMat layout (util.h:390-413):
 *     index[d1+1], item[nnz] (SparseMat::rows), val[nnz] (SparseMat::vals)
 *   - return value 0 = success, negative = error; pcr_last_error() has the text
 *   - functions marked [host] never touch the GPU
 *   - functions marked [device] need a gfx950 GPU and FAIL (never fall back to a
 *     CPU path) when none is usable
 */
#ifndef PRIMALCR_H
#define PRIMALCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_OK               0
#define PCR_ERR_ARG         -1
#define PCR_ERR_IO          -2
#define PCR_ERR_NOMEM       -3
#define PCR_ERR_DEVICE      -4   /* no usable GPU / HIP runtime error            */
#define PCR_ERR_COMM        -5   /* RCCL error                                    */
#define PCR_ERR_STATE       -6   /* call order (e.g. obtain_g before comp_m)      */
#define PCR_ERR_UNSUPPORTED -7

/* solver ids: pmf.h:6  enum {CCDR1, PCR, PCRPP} */
#define PCR_SOLVER_PCR    1
#define PCR_SOLVER_PCRPP  2

/* storage type of U, V, m and the CG vectors on the device.  Prefix sums,
 * objectives, dot products and sweep coefficients are ALWAYS accumulated in fp64. */
#define PCR_F32 0
#define PCR_F64 1

/* mirrors class parameter (pmf.h:9-49) for the fields the PCR/PCR++ path reads,
 * plus device-side extensions */
typedef struct pcr_params {
    int solver_type;   /* pmf.h:32  default PCRPP (2)                             */
    int k;             /* pmf.h:33  rank, default 10                              */
    int threads;       /* pmf.h:38  default 4; host threads of loaders only       */
    int maxiter;       /* pmf.h:35  default 10                                    */
    double lambda;     /* pmf.h:37  default 5000                                  */
    int do_predict;    /* pmf.h:45  default 1                                     */
    int verbose;       /* pmf.h:46                                                */
    double stepsize;   /* pmf.h:28  default 1.0 (reset every step, pcrpp.cpp:877) */
    int ndcg_k;        /* pmf.h:29  default 10                                    */
    /* extensions */
    int precision;     /* PCR_F32 (default) | PCR_F64                             */
    int device;        /* HIP device ordinal, default 0                           */
    /* truncated-Newton knobs: the reference hard-codes 10 CG iterations at
     * most and a residual tolerance of 1 % of ||g|| for both half steps
     * (pcrpp.cpp:340,344 / :632,636).  cg_max_iter = r with a small cg_tol turns the U
     * step into an exact Newton step.  The defaults reproduce the reference.   */
    int cg_max_iter;   /* default 10                                              */
    double cg_tol;     /* default 0.01                                            */
} pcr_params;

/* pmf.h:27-48 parameter::parameter() */
void pcr_params_default(pcr_params *p);                                   /* [host] */

const char *pcr_last_error(void);                                         /* [host] */
const char *pcr_version(void);                                            /* [host] */

/* ------------------------------------------------------------------------- */
/* host data path                                                            */
/* ------------------------------------------------------------------------- */

/* util.cpp:80-93 initial(): N(0,1) from a default-seeded std::default_random_engine
 * through std::normal_distribution<double>; a fresh engine per call.  The same values bit for bit; above 4 M of them
 * produced by several host threads (the engine is an LCG: a range of tries starts at a state that modular
 * exponentiation gives directly, and libstdc++'s polar method takes exactly four draws per try). */
int pcr_initial(double *X, int64_t n, int64_t k);                         /* [host] */
/* rows [row0, row0 + nrows) of what pcr_initial(., n, k) would produce, into X (nrows x k): where an output lands
 * depends on how many tries were accepted before it, so the tries before row0 are still evaluated (in parallel, without
 * their sqrt / log) -- for a rank that holds only its own users' rows of a large U. */
int pcr_initial_rows(double *X, int64_t n, int64_t k, int64_t row0, int64_t nrows);   /* [host] */

typedef struct pcr_dataset pcr_dataset;   /* training CSR + test CSR, host memory */

/* util.cpp:6-25 load() + util.h:197-271 smat_t::load + util.h:360-371
 * testset_t::load + util.cpp:219-274 convert(): reads <dir>/meta and the rating
 * files it names. */
int pcr_dataset_load(const char *dir, pcr_dataset **out);                 /* [host] */
/* the same with `threads` host threads (the CLI passes -n; 0 = up to 16) for every stage: the rating files are mapped, cut at
 * line boundaries and parsed piece by piece into the CSR's own arrays; a file ordered by (user, item) -- what the reference's
 * data sets are -- needs no further data movement, any other order goes through a bucketed counting sort; the test set's
 * rows follow the reference's scan (util.cpp:250-274: the row of the largest user id seen so far, ending at the first
 * id that is no user). */
int pcr_dataset_load_mt(const char *dir, int threads, pcr_dataset **out); /* [host] */
/* Binary side-car of a loaded data set (the reference re-parses the text with fgets/sscanf and re-sorts on
 * every run, util.cpp:6-25, util.h:197-271).  pcr_dataset_save_cache writes the converted CSRs (train + test) to
 * `path`; pcr_dataset_load_cache reads them back (PCR_ERR_IO if the file is missing, truncated or of another version).
 * pcr_dataset_load_cached(dir, threads, cache, out) = load_cache(cache) when the cache exists AND records the same
 * size and modification time of <dir>/meta and of the rating files it names, else load_mt(dir) followed by a best-effort
 * save_cache(cache) (an unwritable cache path is not an error).  The text formats stay the only input format. */
int pcr_dataset_save_cache(const pcr_dataset *ds, const char *path);      /* [host] */
int pcr_dataset_load_cache(const char *path, pcr_dataset **out);          /* [host] */
int pcr_dataset_load_cached(const char *dir, int threads, const char *cache, pcr_dataset **out);   /* [host] */
/* same conversion from in-memory 0-based triplets (train in any order; test must
 * be user-sorted, util.cpp:259-261).  tnnz may be 0. */
int pcr_dataset_from_triplets(int64_t d1, int64_t d2,
                              int64_t nnz, const int32_t *user, const int32_t *item, const double *val,
                              int64_t tnnz, const int32_t *tuser, const int32_t *titem, const double *tval,
                              pcr_dataset **out);                          /* [host] */
/* the same from arrays already in the reference's SparseMat layout (util.h:390-413, what convert() leaves,
 * util.cpp:219-274): index[d1+1], item[nnz] ascending inside a user, val[nnz]; the test CSR (tindex may be NULL) is
 * taken as given. */
int pcr_dataset_from_csr(int64_t d1, int64_t d2, const int64_t *index, const int32_t *item, const double *val,
                         const int64_t *tindex, const int32_t *titem, const double *tval, pcr_dataset **out);   /* [host] */
void pcr_dataset_free(pcr_dataset *ds);                                    /* [host] */
/* sizes: d1, d2, nnz (train), tnnz (test entries assigned by convert()) */
int pcr_dataset_dims(const pcr_dataset *ds, int64_t *d1, int64_t *d2, int64_t *nnz, int64_t *tnnz);
/* copy the CSR out (which: 0 = train, 1 = test); any pointer may be NULL */
int pcr_dataset_csr(const pcr_dataset *ds, int which, int64_t *index, int64_t *item, double *val);
/* #Omega = #{(i,j,k): R_ij > R_ik} after the solver's level bucketing (lround for
 * PrimalCR++, pcrpp.cpp:41; raw doubles for PrimalCR, pcr.cpp:23) */
int64_t pcr_dataset_count_pairs(const pcr_dataset *ds, int solver_type);   /* [host] */

/* A rating file on its own (omp-pmf-predict's test file: pmf-predict.cpp:52-55 reads "user item rating" triples until fscanf
 * fails, no meta file).  count: the file's non-blank lines.  read: at most n entries by `threads` host threads (0 = up to 16),
 * ids 0-based, val may be NULL; the input ends at the first malformed entry: *n_read entries stand before it (the reference's
 * loop never ends on such a line: it tests `!= EOF`, pmf-predict.cpp:56). */
int pcr_rating_file_count(const char *path, int64_t *n);                   /* [host] */
int pcr_rating_file_read(const char *path, int threads, int64_t n, int32_t *user, int32_t *item, double *val,
                         int64_t *n_read);                                 /* [host] */

/* pmf-train.cpp:297-310 + util.cpp:30-51 save_mat_t(U^T,false); save_mat_t(V^T,false):
 * "long d1, long k, d1*k doubles, long d2, long k, d2*k doubles" */
int pcr_model_save(const char *path, const double *U, int64_t d1, const double *V, int64_t d2, int64_t k);
/* pmf-predict.cpp:49-50 + util.cpp:56-79 load_mat_t(fp,true) twice.
 * Call with U = V = NULL to query the sizes first. */
int pcr_model_load(const char *path, int64_t *d1, int64_t *d2, int64_t *k, double *U, double *V);

/* nnz-balanced contiguous user ranges for nparts GPUs: bounds[nparts+1] */
int pcr_partition_users(const int64_t *index, int64_t d1, int nparts, int64_t *bounds);   /* [host] */

/* ------------------------------------------------------------------------- */
/* device solver                                                             */
/* ------------------------------------------------------------------------- */

/* Launch knobs.  The reference has none (its only scheduling choice is `#pragma omp ... schedule(dynamic,500)`,
 * pcrpp.cpp:825); the device solver chooses its launch configuration from the shape of the shard, and these
 * key/value pairs override single choices -- for the parity tests (every configuration must give the same trajectory)
 * and for A/B measurements.  The table belongs to the CALLING THREAD: pcr_solver_create snapshots the creating thread's
 * pairs into the solver (nothing is read later, nothing is shared between threads, so solver handles stay re-entrant);
 * value NULL removes a key; unknown keys are PCR_ERR_ARG.
 * No environment variable is read anywhere on the product path.  (The ROCm runtime has one prerequisite of its own for
 * pcr_solver_comm_init_p2p / RCCL across processes on hosts whose driver only supports dmabuf IPC:
 * HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every rank.)
 *   ustep_mode      1 = latency form of k_ustep for every long class, 2 = throughput form (default: by user count, more than CUs/4
 *                   users = throughput)
 *   cluster_k       4 (default) or 1: workgroups per clustered long user;  cluster_users: how many users get clusters
 *   ubins           "cap:block:resident,..." length classes of the U step below 1024 ratings
 *   ustep_newton    1 = EXACT Newton U step: users of at most 1024 ratings (ranks up to 112) get their direction from the
 *                   explicit r x r Hessian, built on the matrix cores and factored by Cholesky (k_unewton); every other user's CG runs
 *                   to convergence.  Leaves the reference's truncated-CG trajectory on purpose; default 0
 *   vblock_users    n > 0 = BLOCKED-USER V step: the n users with the most ratings (those that rate at least a sixteenth
 *                   of the items) take their share of every Hessian-vector product's two rating-parallel products as dense GEMMs on
 *                   the matrix cores (k_vblock_b, k_vblock_hp) and stay out of the sparse kernels' plan; same results to summation-
 *                   order rounding; default 0
 *   ustep_gram      dual (Gram-matrix on MFMA) U step for users with at most that many ratings (<= 128; default 0 = off)
 *   spmm_tiles, spmm_chunk, sddmm_csc   tiling of the rating-parallel kernels (user tiles per XCD group, ratings per lane group, the
 *                   CG's SDDMM over the tile-major CSC: chosen from the shard's shape)
 *   window_cache    0 = sweeps search their hinge windows instead of caching them (the form users with more than 9 levels take)
 *   prepare_merged  1 / 0 = both LDS classes of k_prepare in one launch, or one launch per length class side by side
 *                   (default: merged below 4 M ratings per shard)
 *   resort_window   half-width D of the nearly-sorted fast path of the per-user sorts (k_prepare, k_ustep's line search): a user
 *                   whose ratings moved by at most D positions since the previous sorted state is re-sorted by windowed rank
 *                   counting (verified; else the full bitonic network); default 8, 0 = always the full network, at most 64
 *   lanes           concurrent streams for length classes (1 = none: the layout a process ends up with when every side stream
 *                   shares a hardware queue with the solver's);  pipeline: 0 = host round trip after every U step
 *   allreduce_chunks N = item ranges of the SpMM whose all-reduces overlap the next range's SpMM (default 1: one all-reduce per
 *                   vector on the solver's stream; opt-in, meant for vectors of 16 MB and more -- about one range per 4 MB)
 *   p2p_ll          peer-to-peer communicator: vectors of at most this many MB (and the objective's scalars) take the device-driven
 *                   exchange (one kernel per rank, flags inside the 8-byte words, no host barrier); larger ones the host-synchronised
 *                   reduce-scatter / all-gather; default 16, 0 = host-synchronised always.  (If any rank cannot allocate
 *                   fine-grained device memory for its exchange boxes, EVERY rank takes the host-synchronised path.)
 *   p2p_timeout_ms  wall-clock deadline of one device-driven exchange (default 20000): a rank whose peers do not arrive in time
 *                   reports PCR_ERR_COMM, poisons its answers (no rank consumes a made-up sum) and raises the job's error flag
 *   p2p_queue_budget   hardware queues one device maps at once for all its processes (default 24: gfx950 under the kernel driver's
 *                   scheduler; 8 of them are left to processes this job cannot see).  Only matters when several ranks SHARE a device
 *                   (a rehearsal): every rank publishes the queues its process holds; past the budget kernels of different processes
 *                   no longer run side by side (0.5 us -> 7-11 ms per dependent hand-off, tools/ubench/queue_budget_probe.hip), so
 *                   the device-driven exchange is switched off for the whole job (host-synchronised exchange, one line on stderr)
 *   count_rows      1 = the U-step kernels count the rows of V they gather (pcr_solver_counter; a diagnostic that
 *                   costs the short-user classes 10-20 %, so off by default)
 *   debug           1 = print launch decisions to stderr
 *   plan_key64      test hook: 1 = the set-up's (tile, item) sort keys in 64 bits -- the form an item side beyond 2^32 / tiles takes anyway
 *   win16, ustep_win_lds   test hooks: 0 = the forms shards with very long users take anyway -- 32-bit window-cache entries (a user
 *                   of 65536 ratings or more), k_ustep reading the window cache from global memory (a class whose LDS is full) --
 *                   forced on small data so that the fuzz tests cover them
 * (Knobs of experiments that are closed -- stream placements, serial classes, cluster hand-off without fences, window-cache
 * widths and copies, sweep load depth, clusters of two -- are gone with their code paths; NOTES.md keeps the numbers.)
 *   fault_cluster_member   test hook: one member of every workgroup cluster leaves early (the launch must report
 *                   PCR_ERR_DEVICE through the bounded hand-off wait instead of hanging)
 *   fault_p2p_skip  test hook: this rank never launches its n-th device-driven exchange (its peers must time out, poison their
 *                   answers and fail with PCR_ERR_COMM; nobody may hang or consume garbage)
 *   fault_p2p_coarse   test hook: this rank behaves as if fine-grained memory were unavailable (all ranks must fall back to the
 *                   host-synchronised exchange together)
 * Stream layout: the solver creates its stream and a high-priority stream, then creates side streams one by one and MEASURES
 * (a few ms each at creation; pcr_tune("debug") prints it) which of them share a hardware queue with the solver's stream (they are
 * not used as lanes; it stops at three lanes, usually after four or five streams) and which lanes share its command-processor PIPE (queues on one pipe share workgroup dispatch and throttle each
 * other): such a lane is placed last, and a high-priority stream that landed on the solver's pipe is replaced.  So the layout
 * no longer depends on how many streams the host application created before the solver; results never did. */
int pcr_tune(const char *key, const char *value);                          /* [host] */

typedef struct pcr_solver pcr_solver;

/* Optional: initialise the HIP runtime for `device` and load the library's code object now (otherwise the first
 * pcr_solver_create does both, ~0.2-0.3 s).  Thread-safe with the [host] functions: a host application can call it on a
 * second thread while it parses its input (omp-pmf-train does: the reference has nothing to overlap, pmf-train.cpp:247-266). */
int pcr_device_warmup(int device);                                         /* [device] */

/* Upload this rank's user shard (rank 0 of 1 = everything) and allocate the
 * device state.  Replaces convert(R), convert(T) at pcrpp.cpp:850-851. */
int pcr_solver_create(const pcr_dataset *ds, const pcr_params *p, int rank, int nranks,
                      pcr_solver **out);                                   /* [device] */
/* The same for a job whose rating set no single process holds (configs[4]: 700 M ratings over 8 GPUs): `ds_local` contains
 * ONLY this rank's users, renumbered from 0 -- users [first_user, first_user + d1(ds_local)) of a job with d1_total users; the
 * host application chooses the ranges (pcr_partition_users on the per-user counts) so that they tile [0, d1_total) in rank
 * order.  Its test set must be empty on every rank or on none.  Replaces the same call sites as pcr_solver_create;
 * the user loop being sharded is pcrpp.cpp:825-833. */
int pcr_solver_create_shard(const pcr_dataset *ds_local, const pcr_params *p, int rank, int nranks,
                            int64_t first_user, int64_t d1_total, pcr_solver **out);   /* [device] */
void pcr_solver_destroy(pcr_solver *s);

/* RCCL bootstrap for nranks > 1 (one process per GPU): rank 0 obtains an id
 * (128 bytes), the host application broadcasts it, every rank calls comm_init. */
int pcr_comm_unique_id(void *id128);                                       /* [device] */
int pcr_solver_comm_init(pcr_solver *s, const void *id128);                /* [device] */
/* The direct peer-to-peer alternative for the ranks of ONE node (replaces the omp atomics of
 * pcrpp.cpp:240-243, :323-327 across GPUs): every rank exposes an exchange buffer through HIP IPC; an all-reduce is a
 * reduce-scatter + all-gather that sums in rank order (every rank obtains the same bits).  Vectors up to 16 MB and the
 * scalars are exchanged by ONE kernel per rank with the flags inside the exchanged 8-byte words (no host involvement);
 * larger vectors through reads of the peers' buffers over xGMI between host barriers over a POSIX shared-memory control
 * block, whose error flag also releases the peers of a rank that failed.  Every rank calls this with the same name ("/something", shm_open);
 * no id exchange is needed.  Use either this or pcr_solver_comm_init. */
int pcr_solver_comm_init_p2p(pcr_solver *s, const char *shm_name);         /* [device] */
/* ranks the solver's communicator reports (ncclCommCount / the p2p control block); 1 without a communicator */
int pcr_solver_comm_nranks(pcr_solver *s);
/* diagnostic counters, cumulative since the solver was created:
 *   "ustep_row_gathers"  rows of V the U steps gathered (per user: 1 for the gradient + 2 per CG iteration + 1 per
 *                        line-search try, times its rating count; all ranks; counted only under pcr_tune("count_rows")) --
 *                        the U step's gather rate = this x k x sizeof(storage type) / its wall time */
/*   "ustep_row_gathers/<slot>"  the same count for ONE length class of the U step on THIS rank (<slot> = its profile slot
 *                        name, e.g. "ustep/256.512"; pcr_solver_ustep_classes lists them, comma-separated) */
int pcr_solver_counter(pcr_solver *s, const char *name, double *value);
int pcr_solver_ustep_classes(pcr_solver *s, char *buf, int64_t cap);
/* Where pcr_solver_create's wall time went -- the unit the reference spends in convert() (util.cpp:219-274) and this library in
 * uploads, the set-up built on the device and the stream probes: phase i (0, 1, ...) of the creation, in order; *name points
 * into the solver (valid until it is destroyed).  PCR_ERR_ARG past the last phase.  omp-pmf-train --timing prints the list. */
int pcr_solver_setup_phase(const pcr_solver *s, int i, const char **name, double *ms);
/* Shard-local mode for a solver created with nranks > 1 and no communicator: every collective
 * becomes a no-op, so pcr_obtain_g / pcr_compute_Ha / pcr_objective return THIS SHARD'S PARTIAL
 * (rank 0 carries the lambda term).  Lets a host application combine shards itself, and lets one
 * process verify the sharding of N ranks on a single GPU. */
int pcr_solver_set_local_only(pcr_solver *s, int on);

/* first local user and number of local users of this rank's shard */
int pcr_solver_shard(const pcr_solver *s, int64_t *first_user, int64_t *n_users, int64_t *nnz_local);

/* factors: host fp64 <-> device.  U is the FULL d1 x k matrix; each rank reads /
 * writes only its own rows [first_user, first_user+n_users). V is d2 x k. */
int pcr_solver_set_factors(pcr_solver *s, const double *U, const double *V);
int pcr_solver_get_factors(pcr_solver *s, double *U, double *V);
/* the same with U_local = this rank's n_users x k rows only (what a rank created by pcr_solver_create_shard holds) */
int pcr_solver_set_factors_local(pcr_solver *s, const double *U_local, const double *V);
int pcr_solver_get_factors_local(pcr_solver *s, double *U_local, double *V);

/* pcrpp.cpp:17-35 comp_m_new (pcr.cpp:47 comp_m): m = u_i . v_j for every rating,
 * from the current device U, V; also builds the per-user (level, m)-sorted state
 * the sweeps use.  m_out (local nnz, CSR order) may be NULL. */
int pcr_comp_m(pcr_solver *s, double *m_out);
/* pcrpp.cpp:361-412 objective_new (pcr.cpp:5 objective) at the state of the last
 * pcr_comp_m: all-rank sum incl. lambda/2 (|U|^2 + |V|^2). */
int pcr_objective(pcr_solver *s, double *obj);
/* pcrpp.cpp:140-249 obtain_g_new (pcr.cpp:102 obtain_g): g (d2 x k) */
int pcr_obtain_g(pcr_solver *s, double *g);
/* pcrpp.cpp:252-332 compute_Ha_new (pcr.cpp:167 compute_Ha): a, Ha are d2 x k */
int pcr_compute_Ha(pcr_solver *s, const double *a, double *Ha);
/* pcrpp.cpp:335-358 solve_delta_new (pcr.cpp:248): CG on H delta = g */
int pcr_solve_delta(pcr_solver *s, const double *g, double *delta, int *cg_iters);
/* pcrpp.cpp:415-444 update_V_new (pcr.cpp:279): one Newton step on V.
 * info[0] = CG iterations, info[1] = line-search evaluations, info[2] = accepted */
int pcr_update_V(pcr_solver *s, double *now_obj, int *info);
/* pcrpp.cpp:818-838 update_U_new (pcr.cpp:587): one Newton step per user.
 * info[0] = total CG iterations, info[1] = total line-search evaluations */
int pcr_update_U(pcr_solver *s, double *now_obj, int64_t *info);
/* util.cpp:434-542 compute_pairwise_error_ndcg on the train (which=0) or test
 * (which=1) ratings with the current device factors */
int pcr_evaluate(pcr_solver *s, int which, int ndcg_k, double *pairwise_err, double *ndcg);

/* what the reference prints per iteration (pcrpp.cpp:859-890) */
typedef struct pcr_iter_stats {
    double obj;
    double train_err, train_ndcg, test_err, test_ndcg;
    double seconds;                 /* cumulative, clock scope of pcrpp.cpp:874-881 */
    int64_t cg_v, ls_v, cg_u, ls_u; /* executed inner-iteration counts              */
} pcr_iter_stats;

typedef void (*pcr_log_fn)(void *ctx, const char *line);

/* pcrpp.cpp:841-901 pcrpp() / pcr.cpp:616-704 pcr(): the whole training loop
 * from the current device factors.  Emits the reference's log lines through
 * `log` (NULL = stdout, rank 0 only; a callback is invoked on EVERY rank of a multi-rank job with the
 * same lines).  hist may be NULL, else holds maxiter+1 records. */
int pcr_train(pcr_solver *s, pcr_log_fn log, void *log_ctx, pcr_iter_stats *hist);
/* The body of that loop (pcrpp.cpp:869-895: update_V_new, update_U_new, no evaluation) n times from the current state,
 * with one host round trip per iteration: the U step is queued without waiting for it and its objective is read back
 * together with the next iteration's line search.  out (may be NULL) receives n records: obj, seconds (cumulative
 * device time of the loop), inner-iteration counts.  pcr_train uses it when do_predict == 0 and log == NULL. */
int pcr_iterate(pcr_solver *s, int n, pcr_iter_stats *out);

/* pmf-predict.cpp:56-64: pred[z] = U[user[z]] . V[item[z]] for n (0-based) pairs */
int pcr_predict(const double *U, int64_t d1, const double *V, int64_t d2, int64_t k,
                int64_t n, const int32_t *user, const int32_t *item, double *pred,
                int device);                                               /* [device] */

/* per-kernel device timing (HIP events on the solver's stream, one pair per launch).
 * slot names: "<class>/<workgroup size>[.<length bound>][g][c][l][r][#n]" for the per-user kernels (classes
 * prepare, vgrad, vhv, ustep; g = global-scratch variant, c = workgroup clusters, l = k_ustep's latency form (8 rows
 * in flight), r = one-wave k_ustep class with LDS-resident rows, #n = the n-th kernel symbol
 * of a workgroup form that two length classes of the U step share -- every class is its own symbol in a
 * profiler's per-kernel tables; "vgrad/all", "vhv/all" = both LDS classes in one launch), "wall:<class>" for
 * the fork..join wall time of a class whose length classes run concurrently, and "sddmm", "spmm",
 * "spmm_fin", "cg", "eval", "allreduce".
 * pcr_profile_list writes the comma-separated names of the slots seen so far.
 * pcr_profile_enable(s, n): n = 0 off, 1 time every launch, n > 1 time every n-th launch of each
 * slot (an event pair costs ~3 us of queue time, so sampling keeps the timed region honest);
 * pcr_profile_get returns the summed time and the number of TIMED launches, pcr_profile_launches the number of
 * launches of the slot since the last reset, timed or not (total time of a sampled slot = average x launches);
 * pcr_profile_scope what ONE launch of the slot covers on this rank: the ratings and users of its length
 * class (the whole shard for the rating-/item-parallel kernels), so that a caller can price a launch
 * without mirroring the class layout. */
int pcr_profile_enable(pcr_solver *s, int on);
int pcr_profile_list(pcr_solver *s, char *buf, int64_t cap);
int pcr_profile_get(pcr_solver *s, const char *name, double *total_ms, int64_t *launches);
int pcr_profile_launches(pcr_solver *s, const char *name, int64_t *launches);
int pcr_profile_scope(pcr_solver *s, const char *name, int64_t *ratings, int64_t *users);
int pcr_profile_reset(pcr_solver *s);
/* blocks until the solver's stream is idle */
int pcr_solver_sync(pcr_solver *s);

#ifdef __cplusplus
}
#endif
#endif /* PRIMALCR_H */
