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
//   --snapshot-every N  also write <model>.iter<k> after every N-th outer iteration
//   --cg-iters N / --cg-tol X  truncated-Newton knobs (the reference hard-codes 10 / 0.01); --cg-iters k with a
//                       small --cg-tol makes the U step an exact Newton step
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

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
        "    --cg-tol x : CG residual tolerance relative to ||g|| (default 0.01, the reference's constant)\n");
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

static void die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, pcr_last_error());
    exit(1);
}

int main(int argc, char** argv) {
    pcr_params param;
    pcr_params_default(&param);
    std::string init_model, cache;
    int snapshot_every = 0;
    int i;
    for (i = 1; i < argc; i++) {                       // pmf-train.cpp:36-108
        if (argv[i][0] != '-') break;
        if (!strcmp(argv[i], "--f64")) { param.precision = PCR_F64; continue; }
        if (++i >= argc) exit_with_help();
        if (!strcmp(argv[i - 1], "--device")) { param.device = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--init-model")) { init_model = argv[i]; continue; }
        if (!strcmp(argv[i - 1], "--cache")) { cache = argv[i]; continue; }
        if (!strcmp(argv[i - 1], "--snapshot-every")) { snapshot_every = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--cg-iters")) { param.cg_max_iter = atoi(argv[i]); continue; }
        if (!strcmp(argv[i - 1], "--cg-tol")) { param.cg_tol = atof(argv[i]); continue; }
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

    pcr_dataset* ds = nullptr;
    if ((cache.empty() ? pcr_dataset_load_mt(input.c_str(), param.threads, &ds)
                       : pcr_dataset_load_cached(input.c_str(), param.threads, cache.c_str(), &ds)) != PCR_OK) die("load");
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
    std::cout << "the rank is " << k << std::endl;
    std::cout << "the number of rows is " << d1 << " and the number of cols is " << d2 << std::endl;
    if (param.solver_type == PCR_SOLVER_PCRPP) { std::cout << nnz << std::endl; std::cout << "starts!" << std::endl; }
    else std::cout << "nnz: " << nnz << std::endl;

    auto t0 = std::chrono::steady_clock::now();
    pcr_solver* s = nullptr;
    if (pcr_solver_create(ds, &param, 0, 1, &s) != PCR_OK) die("solver");
    if (pcr_solver_set_factors(s, U.data(), V.data()) != PCR_OK) die("set_factors");
    SnapCtx snap{s, snapshot_every, model, d1, d2, k, &U, &V, false};
    if (pcr_train(s, snapshot_every > 0 ? snap_log : nullptr, &snap, nullptr) != PCR_OK) die("train");
    if (snap.failed) return 1;
    if (pcr_solver_get_factors(s, U.data(), V.data()) != PCR_OK) die("get_factors");
    printf("Wall-time: %lg secs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

    // side files (pmf-train.cpp:208-227, 276-295)
    std::string suffix = param.solver_type == PCR_SOLVER_PCR ? std::to_string(static_cast<int>(param.lambda)) : "";
    auto dump = [&](const char* name, const std::vector<double>& M, int64_t rows) {
        std::cout << name << " matrix of size " << rows << ", " << k << std::endl;
        std::ofstream f(std::string(name) + suffix + ".txt");
        for (int64_t a = 0; a < rows; ++a)
            for (int b = 0; b < k; ++b) { f << M[a * k + b]; f << (b < k - 1 ? " " : "\n"); }
    };
    dump("U", U, d1);
    dump("V", V, d2);
    if (pcr_model_save(model.c_str(), U.data(), d1, V.data(), d2, k) != PCR_OK) die("model");
    pcr_solver_destroy(s);
    pcr_dataset_free(ds);
    return 0;
}
