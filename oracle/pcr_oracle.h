/*
 * pcr_oracle.h -- CPU ORACLE for the PrimalCR / PrimalCR++ hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The shipped library (primalcr_amd/lib/libprimalcr.so) never links,
 * loads or calls anything in oracle/.
 *
 * What it is: a plain-C, fp64, single-threaded restatement of the reference's
 * algorithm (wuliwei9278/primalCR: pcrpp.cpp, pcr.cpp, util.cpp), written on
 * flat arrays and following the reference's loop order so that results agree
 * with the compiled reference to rounding.  Every function cites the
 * reference file:line it follows.
 *
 * Parity pin: the restatement is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by the UNMODIFIED reference compiled from
 * /root/reference into oracle/_ref/ (recipe: oracle/Makefile, generator:
 * oracle/make_golden.py, vectors: tests/golden/).
 *
 * Data layout (all row-major, 0-based):
 *   U[d1*r], V[d2*r]           factors; row = one user's / item's vector
 *   idx[d1+1]                  user-major CSR row pointer   (SparseMat::index)
 *   item[nnz]                  item id of each rating       (SparseMat::rows)
 *   val[nnz]                   rating                       (SparseMat::vals)
 * The reference's SparseMat::cols (user id per rating) is implied by idx.
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* util.cpp:80-93  initial(): N(0,1) draws from a default-seeded
 * std::default_random_engine (= minstd_rand0) through libstdc++'s
 * std::normal_distribution<double> (Marsaglia polar).  A fresh engine per
 * call, so U and V share one stream (quirk q1). */
void orc_initial(double *X, long n, long k);

/* util.h:201-271 + util.cpp:219-247  smat_t::load_from_iterator + convert():
 * triplets (0-based user,item) in any order -> user-major CSR with items
 * ascending inside a user. Returns 0 on success. */
int orc_build_csr(long d1, long nnz, const int *tu, const int *ti,
                  const double *tv, long *idx, long *item, double *val);

/* util.cpp:250-274  convert(testset_t&): test triplets MUST be user-sorted;
 * the scan breaks at the first entry whose user id exceeds the cursor, which
 * can leave trailing entries unassigned -- mirrored. Returns idx[d1]. */
long orc_build_csr_test(long d1, long nnz, const int *tu, const int *ti,
                        const double *tv, long *idx, long *item, double *val);

/* pcrpp.cpp:17-35 */
void orc_comp_m(const double *U, const double *V, long d1, const long *idx,
                const long *item, int r, double *m);
/* pcrpp.cpp:361-412 */
double orc_objective_new(const double *m, const double *U, const double *V,
                         long d1, long d2, const long *idx, const double *val,
                         int r, double lambda);
/* pcrpp.cpp:140-249   g is d2*r */
void orc_obtain_g_new(const double *U, const double *V, long d1, long d2,
                      const long *idx, const long *item, const double *val,
                      const double *m, int r, double lambda, double *g);
/* pcrpp.cpp:252-332   a, Ha are d2*r */
void orc_compute_Ha_new(const double *a, const double *m, const double *U,
                        long d1, long d2, const long *idx, const long *item,
                        const double *val, int r, double lambda, double *Ha);
/* pcrpp.cpp:335-358   returns number of CG iterations executed */
int orc_solve_delta_new(const double *g, const double *m, const double *U,
                        long d1, long d2, const long *idx, const long *item,
                        const double *val, int r, double lambda,
                        double *delta);
/* pcrpp.cpp:415-444   V in/out; m_out[nnz] = m of the LAST TRIED V_new;
 * returns number of line-search evaluations (1..20); *accepted = 1 if a step
 * was accepted. */
int orc_update_V_new(long d1, long d2, const long *idx, const long *item,
                     const double *val, double lambda, double stepsize, int r,
                     const double *U, double *V, double *now_obj,
                     double *m_out, int *accepted, int *cg_iters);
/* pcrpp.cpp:779-815   one user. ui_new[r] out. returns CG iterations (0 if
 * skipped); *n_ls = line-search evaluations. */
int orc_update_u_new(long i, const double *V, const long *idx,
                     const long *item, const double *val, const double *m,
                     int r, double lambda, double stepsize, const double *ui,
                     double *ui_new, double *obj_u_new, int *n_ls);
/* pcrpp.cpp:818-838   U_new is d1*r */
void orc_update_U_new(long d1, long d2, const long *idx, const long *item,
                      const double *val, const double *m, double lambda,
                      double stepsize, int r, const double *V,
                      const double *U, double *U_new, double *now_obj,
                      long *total_cg, long *total_ls);

/* Not in the reference (it hard-codes 10 / 0.01): moves the CG cap and tolerance of every
 * solve above, to check the product's cg_max_iter / cg_tol extension.  Process-global. */
void orc_set_cg(int max_iter, double tol);

/* util.cpp:434-542 */
void orc_eval(const double *U, const double *V, long d1, const long *idx,
              const long *item, const double *val, int r, int ndcg_k,
              double *pairwise_err, double *ndcg);

/* ---- PrimalCR (solver 1), pcr.cpp ---- */
/* pcr.cpp:5-43 */
double orc_objective(const double *m, const double *U, const double *V,
                     long d1, long d2, const long *idx, const double *val,
                     int r, double lambda);
/* pcr.cpp:102-164 */
void orc_obtain_g(const double *U, const double *V, long d1, long d2,
                  const long *idx, const long *item, const double *val,
                  const double *m, int r, double lambda, double *g);
/* pcr.cpp:167-243 */
void orc_compute_Ha(const double *a, const double *m, const double *U,
                    long d1, long d2, const long *idx, const long *item,
                    const double *val, int r, double lambda, double *Ha);
/* pcr.cpp:523-585 */
int orc_update_u(long i, const double *V, const long *idx, const long *item,
                 const double *val, const double *m, int r, double lambda,
                 double stepsize, const double *ui, double *ui_new,
                 double *obj_u_new, int *n_ls);

/* Per-iteration record of a training run (what the reference prints). */
typedef struct {
    double obj;          /* "Iter k ... obj"                        */
    double train_err, train_ndcg;   /* valid if do_predict           */
    double test_err, test_ndcg;     /* valid if do_predict && tnnz   */
    double seconds;      /* cumulative, same clock scope as pcrpp.cpp:874-881 */
    long cg_v, ls_v, cg_u, ls_u;    /* executed inner-iteration counts */
} orc_iter_t;

/* pcrpp.cpp:841-901 (solver=2) / pcr.cpp:616-704 (solver=1).
 * U, V in/out. hist must hold maxiter+1 records (record 0 = "Iter 0").
 * test arrays may be NULL (tnnz = 0). */
void orc_train(int solver, long d1, long d2, const long *idx,
               const long *item, const double *val, const long *tidx,
               const long *titem, const double *tval, long tnnz, int r,
               double lambda, int maxiter, int do_predict, int ndcg_k,
               double stepsize, double *U, double *V, orc_iter_t *hist);

/* number of ordered pairs (i,j,k) with R_ij > R_ik after lround bucketing
 * (= the objective at U=V=0).  solver1=1 counts on raw doubles. */
long orc_count_pairs(long d1, const long *idx, const double *val, int raw);

#ifdef __cplusplus
}
#endif
#endif
