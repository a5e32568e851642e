// omp-pmf-predict -- drop-in replacement of pmf-predict.cpp: loads a model file, scores every
// "user item rating" line of the test file on the GPU (batched SDDMM) and writes one "%lf\n" per
// line (pmf-predict.cpp:15-67).  Unlike the reference, out-of-range ids are reported, not UB.
//   omp-pmf-predict --host test_file model output_file
// scores on the host instead -- the reference's own loop (one fp64 dot product per line, pmf-predict.cpp:56-64), for a machine
// without a GPU (BASELINE configs[0]: "runs without a GPU").  Only on request: without --host a missing device is an error.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "primalcr.h"

int main(int argc, char** argv) {
    bool on_host = false;
    if (argc >= 2 && !strcmp(argv[1], "--host")) { on_host = true; ++argv; --argc; }
    if (argc != 4) { printf("Usage: omp-pmf-predict test_file model output_file\n"); return 1; }
    FILE* test_fp = fopen(argv[1], "r");
    if (!test_fp) { fprintf(stderr, "can't open test file %s\n", argv[1]); return 1; }
    FILE* out_fp = fopen(argv[3], "wb");
    if (!out_fp) { fprintf(stderr, "can't open output file %s\n", argv[3]); return 1; }
    int64_t d1, d2, k;
    if (pcr_model_load(argv[2], &d1, &d2, &k, nullptr, nullptr) != PCR_OK) { fprintf(stderr, "can't open model file %s\n", argv[2]); return 1; }
    std::vector<double> U((size_t)d1 * k), V((size_t)d2 * k);
    if (pcr_model_load(argv[2], &d1, &d2, &k, U.data(), V.data()) != PCR_OK) { fprintf(stderr, "%s\n", pcr_last_error()); return 1; }
    std::vector<int32_t> user, item;
    int i, j;
    double v;
    while (fscanf(test_fp, "%d %d %lf", &i, &j, &v) == 3) { user.push_back(i - 1); item.push_back(j - 1); }
    fclose(test_fp);
    std::vector<double> pred(user.size());
    if (on_host) {
        for (size_t z = 0; z < user.size(); ++z) {
            if (user[z] < 0 || user[z] >= d1 || item[z] < 0 || item[z] >= d2) { fprintf(stderr, "predict: pair %zu outside the model\n", z); return 1; }
            const double *u = U.data() + (size_t)user[z] * k, *w = V.data() + (size_t)item[z] * k;
            double dot = 0.0;
            for (int64_t t = 0; t < k; ++t) dot += u[t] * w[t];          // pmf-predict.cpp:58-62
            pred[z] = dot;
        }
    } else if (pcr_predict(U.data(), d1, V.data(), d2, k, (int64_t)user.size(), user.data(), item.data(), pred.data(), 0) != PCR_OK) {
        fprintf(stderr, "predict: %s\n", pcr_last_error());
        return 1;
    }
    for (double p : pred) fprintf(out_fp, "%lf\n", p);
    fclose(out_fp);
    return 0;
}
