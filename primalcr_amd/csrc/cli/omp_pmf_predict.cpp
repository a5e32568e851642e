// omp-pmf-predict -- drop-in replacement of pmf-predict.cpp: loads a model file, scores every
// "user item rating" line of the test file on the GPU (batched SDDMM) and writes one "%lf\n" per
// line (pmf-predict.cpp:15-67).  Unlike the reference, out-of-range ids are reported, not UB.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "primalcr.h"

int main(int argc, char** argv) {
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
    if (pcr_predict(U.data(), d1, V.data(), d2, k, (int64_t)user.size(), user.data(), item.data(), pred.data(), 0) != PCR_OK) {
        fprintf(stderr, "predict: %s\n", pcr_last_error());
        return 1;
    }
    for (double p : pred) fprintf(out_fp, "%lf\n", p);
    fclose(out_fp);
    return 0;
}
