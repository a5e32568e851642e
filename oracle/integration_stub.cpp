// oracle/integration_stub.cpp -- TEST INFRASTRUCTURE (link test of the drop-in boundary; nothing under primalcr_amd/ uses it).
//
// The reference-side binding of INTEGRATION.md section 1 as a real program: compiled against the reference's OWN headers
// (-I$(REF): pmf.h, util.h where they lie under /root/reference), linked with the reference's OWN objects
// (oracle/_ref/util.o pcr.o pcrpp.o ccd-r1.o, built by oracle/Makefile from the unmodified sources) and with
// libprimalcr.so.  main() follows run_pcrpp() (pmf-train.cpp:247-314): the reference's load(), the reference's initial(),
// then the solver call -- pcrpp(X, U, V, T, param) at pmf-train.cpp:273 -- replaced by pcrpp_mi355x(), and the reference's
// save_mat_t() for the model file.  Its stdout is compared with the unmodified binary's on the golden data sets
// (tests/test_cli.py).  No reference source text is copied: the types and functions used are declared by the reference's
// headers.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "pmf.h"             // the reference's: class parameter, smat_t, testset_t, mat_t, load, initial, save_mat_t
#include "primalcr.h"        // this repository's C ABI

static int g_precision = PCR_F32;

// ---- INTEGRATION.md section 1, verbatim ------------------------------------------------------------------------------------
static void pcrpp_mi355x(smat_t &R, mat_t &U, mat_t &V, testset_t &T, parameter &param) {
    // smat_t is CSC over items with row_idx = user (util.h:157-166): hand the triplets over,
    // the library rebuilds the user-major CSR exactly like convert() (util.cpp:219-247)
    std::vector<int32_t> user(R.nnz), item(R.nnz);
    for (long c = 0; c < R.cols; ++c)
        for (long idx = R.col_ptr[c]; idx < R.col_ptr[c + 1]; ++idx) { user[idx] = R.row_idx[idx]; item[idx] = c; }
    std::vector<int32_t> tu(T.nnz), ti(T.nnz); std::vector<double> tv(T.nnz);
    for (long z = 0; z < T.nnz; ++z) { tu[z] = T.T[z].i; ti[z] = T.T[z].j; tv[z] = T.T[z].v; }
    pcr_dataset *ds; pcr_solver *s;
    if (pcr_dataset_from_triplets(R.rows, R.cols, R.nnz, user.data(), item.data(), R.val,
                                  T.nnz, tu.data(), ti.data(), tv.data(), &ds)) { fprintf(stderr, "%s\n", pcr_last_error()); exit(1); }
    pcr_params p; pcr_params_default(&p);
    p.solver_type = param.solver_type; p.k = param.k; p.threads = param.threads; p.maxiter = param.maxiter;
    p.lambda = param.lambda; p.do_predict = param.do_predict; p.stepsize = param.stepsize; p.ndcg_k = param.ndcg_k;
    p.precision = g_precision;
    const int k = param.k;
    std::vector<double> Uf(U.size() * k), Vf(V.size() * k);           // mat_t -> flat row-major
    for (size_t i = 0; i < U.size(); ++i) std::copy(U[i].begin(), U[i].end(), Uf.begin() + i * k);
    for (size_t i = 0; i < V.size(); ++i) std::copy(V[i].begin(), V[i].end(), Vf.begin() + i * k);
    if (pcr_solver_create(ds, &p, /*rank*/0, /*nranks*/1, &s) ||
        pcr_solver_set_factors(s, Uf.data(), Vf.data()) ||
        pcr_train(s, /*log*/NULL, NULL, /*hist*/NULL) ||             // prints the same lines as pcrpp()
        pcr_solver_get_factors(s, Uf.data(), Vf.data())) { fprintf(stderr, "%s\n", pcr_last_error()); exit(1); }
    for (size_t i = 0; i < U.size(); ++i) std::copy(Uf.begin() + i * k, Uf.begin() + (i + 1) * k, U[i].begin());
    for (size_t i = 0; i < V.size(); ++i) std::copy(Vf.begin() + i * k, Vf.begin() + (i + 1) * k, V[i].begin());
    pcr_solver_destroy(s); pcr_dataset_free(ds);
}
// -----------------------------------------------------------------------------------------------------------------------------

int main(int argc, char **argv) {
    parameter param;                                     // the reference's defaults (pmf.h:27-48)
    param.solver_type = 2;
    int i = 1;
    for (; i < argc && argv[i][0] == '-'; ++i) {
        if (!strcmp(argv[i], "--f64")) { g_precision = PCR_F64; continue; }
        if (i + 1 >= argc) { fprintf(stderr, "option %s needs a value\n", argv[i]); return 1; }
        switch (argv[i][1]) {
            case 's': param.solver_type = atoi(argv[++i]); break;
            case 'k': param.k = atoi(argv[++i]); break;
            case 'l': param.lambda = atof(argv[++i]); break;
            case 't': param.maxiter = atoi(argv[++i]); break;
            case 'p': param.do_predict = atoi(argv[++i]); break;
            default: fprintf(stderr, "unknown option %s\n", argv[i]); return 1;
        }
    }
    if (i + 1 >= argc) { fprintf(stderr, "usage: %s [-s 1|2] [-k rank] [-l lambda] [-t iters] [-p 0|1] [--f64] data_dir model\n", argv[0]); return 1; }
    smat_t X; testset_t T; mat_t U, V;
    FILE *model_fp = fopen(argv[i + 1], "wb");
    if (!model_fp) { fprintf(stderr, "can't open output file %s\n", argv[i + 1]); return 1; }
    load(argv[i], X, T, false);                          // the reference's loader (util.cpp:6-25)
    initial(U, X.rows, param.k);                         // the reference's init (util.cpp:80-93)
    initial(V, X.cols, param.k);
    std::cout << "the rank is " << param.k << std::endl;
    std::cout << "the number of rows is " << X.rows << " and the number of cols is " << X.cols << std::endl;
    pcrpp_mi355x(X, U, V, T, param);                     // <- pmf-train.cpp:273 (pcrpp) / :204 (pcr)
    mat_t UT(param.k, vec_t(U.size())), VT(param.k, vec_t(V.size()));
    for (size_t a = 0; a < U.size(); ++a) for (int b = 0; b < param.k; ++b) UT[b][a] = U[a][b];
    for (size_t a = 0; a < V.size(); ++a) for (int b = 0; b < param.k; ++b) VT[b][a] = V[a][b];
    save_mat_t(UT, model_fp, false);                     // the reference's writer (util.cpp:30-51)
    save_mat_t(VT, model_fp, false);
    fclose(model_fp);
    return 0;
}
