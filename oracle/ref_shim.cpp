// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (oracle/, see pcr_oracle.h).
//
// A thin C-linkage driver around the UNMODIFIED reference objects
// (util.o, pcr.o, pcrpp.o compiled from /root/reference where they lie; see
// oracle/Makefile).  It contains no reference code: only prototypes of the
// reference's externally-linked functions, and flat-array <-> mat_t/SparseMat
// marshalling, so that oracle/make_golden.py can dump per-function golden
// vectors and tests can cross-check the C restatement in this container.
// The resulting oracle/_ref/libpcrref.so is git-ignored and never shipped.
#include "util.h"
#include "pmf.h"

// prototypes of reference functions defined in pcrpp.cpp (no header there)
double* comp_m_new(const mat_t& U, const mat_t& V, SparseMat* X, int r);
mat_t obtain_g_new(const mat_t& U, const mat_t& V, SparseMat* X, double* m, double lambda);
vec_t compute_Ha_new(const vec_t& a, double* m, const mat_t& U, SparseMat* X, int r, double lambda);
vec_t solve_delta_new(const vec_t& g, double* m, const mat_t& U, SparseMat* X, int r, double lambda);
double objective_new(double* m, const mat_t& U, const mat_t& V, SparseMat* X, double lambda);
double* update_V_new(SparseMat* X, double lambda, double stepsize, int r, const mat_t& U, mat_t& V, double& now_obj);
vec_t update_u_new(long i, const mat_t& V, SparseMat* X, double* m, int r, double lambda, double stepsize, const vec_t& ui, double& obj_u_new);
mat_t update_U_new(SparseMat* X, double* m, double lambda, double stepsize, int r, const mat_t& V, const mat_t& U, double& now_obj);
// pcr.cpp
double objective(double* m, const mat_t& U, const mat_t& V, SparseMat* X, double lambda);
mat_t obtain_g(const mat_t& U, const mat_t& V, SparseMat* X, double* m, double lambda);
vec_t compute_Ha(const vec_t& a, double* m, const mat_t& U, SparseMat* X, int r, double lambda);
vec_t update_u(long i, const mat_t& V, SparseMat* X, double* m, int r, double lambda, double stepsize, const vec_t& ui, double& obj_u_new);
double* update_V(SparseMat* X, double lambda, double stepsize, int r, const mat_t& U, mat_t& V, double& now_obj);
mat_t update_U(SparseMat* X, double* m, double lambda, double stepsize, int r, const mat_t& V, const mat_t& U, double& now_obj);

namespace {
mat_t to_mat(const double* A, long rows, int r) {
    mat_t M(rows, vec_t(r));
    for (long i = 0; i < rows; ++i)
        for (int j = 0; j < r; ++j) M[i][j] = A[i * r + j];
    return M;
}
void from_mat(const mat_t& M, double* A) {
    size_t c = 0;
    for (size_t i = 0; i < M.size(); ++i)
        for (size_t j = 0; j < M[i].size(); ++j) A[c++] = M[i][j];
}
SparseMat* to_sp(long d1, long d2, const long* idx, const long* item, const double* val) {
    long nnz = idx[d1];
    SparseMat* X = new SparseMat(d1, d2, nnz);
    for (long i = 0; i <= d1; ++i) X->index[i] = idx[i];
    for (long i = 0; i < d1; ++i)
        for (long z = idx[i]; z < idx[i + 1]; ++z) {
            X->cols[z] = i;
            X->rows[z] = item[z];
            X->vals[z] = val[z];
        }
    return X;
}
}  // namespace

extern "C" {

void ref_set_threads(int n) { omp_set_num_threads(n); }

void ref_initial(double* X, long n, long k) {
    mat_t M;
    initial(M, n, k);
    from_mat(M, X);
}

// load(dir) + convert(R) + convert(T): returns sizes; arrays filled if non-null
int ref_load_dir(const char* dir, long* d1, long* d2, long* nnz, long* tnnz,
                 long* idx, long* item, double* val,
                 long* tidx, long* titem, double* tval) {
    smat_t R;
    testset_t T;
    load(dir, R, T, false);
    *d1 = R.rows; *d2 = R.cols; *nnz = R.nnz; *tnnz = T.nnz;
    if (!idx) return 0;
    SparseMat* X = convert(R);
    SparseMat* XT = convert(T, X->d1, X->d2);
    for (long i = 0; i <= X->d1; ++i) idx[i] = X->index[i];
    for (long z = 0; z < X->nnz; ++z) { item[z] = X->rows[z]; val[z] = X->vals[z]; }
    if (tidx) {
        for (long i = 0; i <= X->d1; ++i) tidx[i] = XT->index[i];
        for (long z = 0; z < XT->index[X->d1]; ++z) { titem[z] = XT->rows[z]; tval[z] = XT->vals[z]; }
    }
    delete X; delete XT;
    return 0;
}

void ref_comp_m_new(const double* U, const double* V, long d1, long d2, const long* idx,
                    const long* item, const double* val, int r, double* m) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    double* mm = comp_m_new(to_mat(U, d1, r), to_mat(V, d2, r), X, r);
    for (long z = 0; z < X->nnz; ++z) m[z] = mm[z];
    delete[] mm; delete X;
}

double ref_objective_new(const double* m, const double* U, const double* V, long d1, long d2,
                         const long* idx, const long* item, const double* val, int r, double lambda) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    double res = objective_new(const_cast<double*>(m), to_mat(U, d1, r), to_mat(V, d2, r), X, lambda);
    delete X;
    return res;
}

void ref_obtain_g_new(const double* U, const double* V, long d1, long d2, const long* idx,
                      const long* item, const double* val, const double* m, int r, double lambda, double* g) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    from_mat(obtain_g_new(to_mat(U, d1, r), to_mat(V, d2, r), X, const_cast<double*>(m), lambda), g);
    delete X;
}

void ref_compute_Ha_new(const double* a, const double* m, const double* U, long d1, long d2,
                        const long* idx, const long* item, const double* val, int r, double lambda, double* Ha) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    vec_t av(a, a + d2 * r);
    vec_t res = compute_Ha_new(av, const_cast<double*>(m), to_mat(U, d1, r), X, r, lambda);
    for (size_t i = 0; i < res.size(); ++i) Ha[i] = res[i];
    delete X;
}

void ref_solve_delta_new(const double* g, const double* m, const double* U, long d1, long d2,
                         const long* idx, const long* item, const double* val, int r, double lambda, double* delta) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    vec_t gv(g, g + d2 * r);
    vec_t res = solve_delta_new(gv, const_cast<double*>(m), to_mat(U, d1, r), X, r, lambda);
    for (size_t i = 0; i < res.size(); ++i) delta[i] = res[i];
    delete X;
}

void ref_update_V_new(long d1, long d2, const long* idx, const long* item, const double* val,
                      double lambda, double stepsize, int r, const double* U, double* V,
                      double* now_obj, double* m_out) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    mat_t Vm = to_mat(V, d2, r);
    double obj = 0.0;
    double* mm = update_V_new(X, lambda, stepsize, r, to_mat(U, d1, r), Vm, obj);
    from_mat(Vm, V);
    *now_obj = obj;
    for (long z = 0; z < X->nnz; ++z) m_out[z] = mm[z];
    delete[] mm; delete X;
}

void ref_update_u_new(long i, const double* V, long d1, long d2, const long* idx, const long* item,
                      const double* val, const double* m, int r, double lambda, double stepsize,
                      const double* ui, double* ui_new, double* obj_u_new) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    vec_t u(ui, ui + r);
    double obj = 0.0;
    vec_t res = update_u_new(i, to_mat(V, d2, r), X, const_cast<double*>(m), r, lambda, stepsize, u, obj);
    for (int t = 0; t < r; ++t) ui_new[t] = res[t];
    *obj_u_new = obj;
    delete X;
}

void ref_update_U_new(long d1, long d2, const long* idx, const long* item, const double* val,
                      const double* m, double lambda, double stepsize, int r, const double* V,
                      const double* U, double* U_new, double* now_obj) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    double obj = 0.0;
    from_mat(update_U_new(X, const_cast<double*>(m), lambda, stepsize, r, to_mat(V, d2, r), to_mat(U, d1, r), obj), U_new);
    *now_obj = obj;
    delete X;
}

void ref_eval(const double* U, const double* V, long d1, long d2, const long* idx, const long* item,
              const double* val, int r, int ndcg_k, double* err, double* ndcg) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    pair<double, double> res = compute_pairwise_error_ndcg(to_mat(U, d1, r), to_mat(V, d2, r), X, ndcg_k);
    *err = res.first; *ndcg = res.second;
    delete X;
}

// ---- solver 1 (pcr.cpp) ----
double ref_objective(const double* m, const double* U, const double* V, long d1, long d2,
                     const long* idx, const long* item, const double* val, int r, double lambda) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    double res = objective(const_cast<double*>(m), to_mat(U, d1, r), to_mat(V, d2, r), X, lambda);
    delete X;
    return res;
}
void ref_obtain_g(const double* U, const double* V, long d1, long d2, const long* idx,
                  const long* item, const double* val, const double* m, int r, double lambda, double* g) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    from_mat(obtain_g(to_mat(U, d1, r), to_mat(V, d2, r), X, const_cast<double*>(m), lambda), g);
    delete X;
}
void ref_compute_Ha(const double* a, const double* m, const double* U, long d1, long d2,
                    const long* idx, const long* item, const double* val, int r, double lambda, double* Ha) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    vec_t av(a, a + d2 * r);
    vec_t res = compute_Ha(av, const_cast<double*>(m), to_mat(U, d1, r), X, r, lambda);
    for (size_t i = 0; i < res.size(); ++i) Ha[i] = res[i];
    delete X;
}
void ref_update_u(long i, const double* V, long d1, long d2, const long* idx, const long* item,
                  const double* val, const double* m, int r, double lambda, double stepsize,
                  const double* ui, double* ui_new, double* obj_u_new) {
    SparseMat* X = to_sp(d1, d2, idx, item, val);
    vec_t u(ui, ui + r);
    double obj = 0.0;
    vec_t res = update_u(i, to_mat(V, d2, r), X, const_cast<double*>(m), r, lambda, stepsize, u, obj);
    for (int t = 0; t < r; ++t) ui_new[t] = res[t];
    *obj_u_new = obj;
    delete X;
}

}  // extern "C"
