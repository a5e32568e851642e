// sanitize/host_tsan_harness.cpp -- ThreadSanitizer run of the multi-threaded host data path (pcr_host.cpp): the loader's pieces
// (line counting, parsing into disjoint slices, the ordered-file fast path and the bucketed counting sort, the test set's running
// maximum), pcr_dataset_from_triplets, the level builder and the parallel pcr_initial -- every stage hands disjoint ranges to its
// worker threads and joins them before the next one reads; TSan checks that claim.  Built by `make tsan` from pcr_host.cpp itself
// (no device code is linked); tests/test_sanitizers.py runs it.  Exit code 0 and no "WARNING: ThreadSanitizer" = clean.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../pcr_host.h"

static int fail(const char* what) { fprintf(stderr, "[host harness] FAILED: %s: %s\n", what, pcr_last_error()); return 1; }

int main() {
    char tmpl[] = "/tmp/pcr_tsan_XXXXXX";
    if (!mkdtemp(tmpl)) return fail("mkdtemp");
    const std::string dir = tmpl;
    const int64_t d1 = 3000, d2 = 900;
    std::mt19937_64 rng(7);
    std::vector<int32_t> user, item;
    std::vector<double> val;
    for (int64_t u = 0; u < d1; ++u) {
        const int n = 20 + (int)(rng() % 60);
        int32_t j = (int32_t)(rng() % 5);
        for (int q = 0; q < n && j < d2; ++q) { user.push_back((int32_t)u); item.push_back(j); val.push_back(1.0 + (double)(rng() % 5)); j += 1 + (int32_t)(rng() % 20); }
    }
    const int64_t nnz = (int64_t)user.size();
    auto write_dir = [&](const std::vector<int64_t>& order, const char* sub) {
        const std::string d = dir + "/" + sub;
        mkdir(d.c_str(), 0700);
        FILE* f = fopen((d + "/training.ratings").c_str(), "w");
        for (int64_t z : order) fprintf(f, "%d %d %d\n", user[z] + 1, item[z] + 1, (int)val[z]);
        fclose(f);
        f = fopen((d + "/test.ratings").c_str(), "w");
        for (int64_t u = 0; u < d1; ++u) fprintf(f, "%ld %d 3\n%ld %d 5\n", (long)(u + 1), 1 + (int)(u % d2), (long)(u > 10 ? u - 7 : u + 1), 2 + (int)(u % (d2 - 2)));
        fclose(f);
        f = fopen((d + "/meta").c_str(), "w");
        fprintf(f, "%ld %ld\n%ld training.ratings\n%ld test.ratings\n", (long)d1, (long)d2, (long)nnz, (long)(2 * d1));
        fclose(f);
        return d;
    };
    std::vector<int64_t> order((size_t)nnz);
    for (int64_t z = 0; z < nnz; ++z) order[(size_t)z] = z;
    const std::string sorted_dir = write_dir(order, "sorted");
    for (int64_t z = nnz - 1; z > 0; --z) std::swap(order[(size_t)z], order[(size_t)(rng() % (uint64_t)(z + 1))]);
    const std::string shuffled_dir = write_dir(order, "shuffled");
    int bad = 0;
    pcr_dataset *a = nullptr, *b = nullptr, *c = nullptr;
    if (pcr_dataset_load_mt(sorted_dir.c_str(), 8, &a) != PCR_OK) return fail("load sorted");
    if (pcr_dataset_load_mt(shuffled_dir.c_str(), 8, &b) != PCR_OK) return fail("load shuffled");
    if (pcr_dataset_from_triplets(d1, d2, nnz, user.data(), item.data(), val.data(), 0, nullptr, nullptr, nullptr, &c) != PCR_OK) return fail("from_triplets");
    for (pcr_dataset* x : {b, c}) {
        if (x->train.index != a->train.index || x->train.item.size() != a->train.item.size() ||
            memcmp(x->train.item.data(), a->train.item.data(), a->train.item.size() * sizeof(int32_t)) != 0 ||
            memcmp(x->train.val.data(), a->train.val.data(), a->train.val.size() * sizeof(double)) != 0) { fprintf(stderr, "[host harness] CSRs differ\n"); bad = 1; }
    }
    if (a->test.index != b->test.index || a->test.nnz() != 2 * d1) { fprintf(stderr, "[host harness] test sets differ (%ld entries)\n", (long)a->test.nnz()); bad = 1; }
    if (pcr_dataset_count_pairs(a, PCR_SOLVER_PCRPP) <= 0 || pcr_dataset_count_pairs(a, PCR_SOLVER_PCR) != pcr_dataset_count_pairs(b, PCR_SOLVER_PCR)) { fprintf(stderr, "[host harness] pair counts differ\n"); bad = 1; }
    pcr_dataset_free(a); pcr_dataset_free(b); pcr_dataset_free(c);
    {   // the parallel initial() against the serial stream (below the threshold the library runs the reference's own loop)
        const int64_t n = 45000, k = 100;                      // 4.5 M values: above the threshold
        std::vector<double> X((size_t)(n * k)), head((size_t)(1000 * k)), rows((size_t)(500 * k));
        if (pcr_initial(X.data(), n, k) != PCR_OK || pcr_initial(head.data(), 1000, k) != PCR_OK || pcr_initial_rows(rows.data(), n, k, 30000, 500) != PCR_OK) return fail("initial");
        if (memcmp(X.data(), head.data(), head.size() * sizeof(double)) != 0 || memcmp(X.data() + 30000 * k, rows.data(), rows.size() * sizeof(double)) != 0) {
            fprintf(stderr, "[host harness] parallel initial() differs from the serial stream\n"); bad = 1;
        }
    }
    std::string cmd = "rm -rf " + dir;
    if (system(cmd.c_str()) != 0) bad = 1;
    fprintf(stderr, bad ? "[host harness] FAILED\n" : "[host harness] loader, CSR build, levels and initial() behaved\n");
    return bad;
}
