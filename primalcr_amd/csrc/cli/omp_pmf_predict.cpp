// omp-pmf-predict -- drop-in replacement of pmf-predict.cpp: loads a model file, scores every
// "user item rating" line of the test file on the GPU (batched SDDMM) and writes one "%lf\n" per
// line (pmf-predict.cpp:15-67).  Unlike the reference, out-of-range ids are reported, not UB.
//   omp-pmf-predict --host test_file model output_file
// scores on the host instead -- the reference's own loop (one fp64 dot product per line, pmf-predict.cpp:56-64), for a machine
// without a GPU (BASELINE configs[0]: "runs without a GPU").  Only on request: without --host a missing device is an error.
//
// The reference reads the test file with one fscanf per line and writes one fprintf per line (pmf-predict.cpp:52-64); here the file
// is mapped and parsed by the host threads (pcr_rating_file_read: the input ends at the first malformed entry -- the reference's
// loop tests `!= EOF`, pmf-predict.cpp:56, and prints the previous pair's score for ever on such a line) and
// the output is formatted by threads too ("%lf" = std::to_chars in its fixed format at precision 6, which the standard defines
// through that printf conversion) -- the same bytes.
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "primalcr.h"

int main(int argc, char** argv) {
    bool on_host = false;
    if (argc >= 2 && !strcmp(argv[1], "--host")) { on_host = true; ++argv; --argc; }
    if (argc != 4) { printf("Usage: omp-pmf-predict test_file model output_file\n"); return 1; }
    FILE* test_fp = fopen(argv[1], "r");
    if (!test_fp) { fprintf(stderr, "can't open test file %s\n", argv[1]); return 1; }
    fclose(test_fp);
    FILE* out_fp = fopen(argv[3], "wb");
    if (!out_fp) { fprintf(stderr, "can't open output file %s\n", argv[3]); return 1; }
    int64_t d1, d2, k;
    if (pcr_model_load(argv[2], &d1, &d2, &k, nullptr, nullptr) != PCR_OK) { fprintf(stderr, "can't open model file %s\n", argv[2]); return 1; }
    std::vector<double> U((size_t)d1 * k), V((size_t)d2 * k);
    if (pcr_model_load(argv[2], &d1, &d2, &k, U.data(), V.data()) != PCR_OK) { fprintf(stderr, "%s\n", pcr_last_error()); return 1; }
    int64_t lines = 0, n = 0;
    if (pcr_rating_file_count(argv[1], &lines) != PCR_OK) { fprintf(stderr, "%s\n", pcr_last_error()); return 1; }
    std::vector<int32_t> user((size_t)lines), item((size_t)lines);
    if (pcr_rating_file_read(argv[1], 0, lines, user.data(), item.data(), nullptr, &n) != PCR_OK) { fprintf(stderr, "%s\n", pcr_last_error()); return 1; }
    std::vector<double> pred((size_t)n);
    if (on_host) {
        for (int64_t z = 0; z < n; ++z) {
            if (user[z] < 0 || user[z] >= d1 || item[z] < 0 || item[z] >= d2) { fprintf(stderr, "predict: pair %ld outside the model\n", (long)z); return 1; }
            const double *u = U.data() + (size_t)user[z] * k, *w = V.data() + (size_t)item[z] * k;
            double dot = 0.0;
            for (int64_t t = 0; t < k; ++t) dot += u[t] * w[t];          // pmf-predict.cpp:58-62
            pred[z] = dot;
        }
    } else if (pcr_predict(U.data(), d1, V.data(), d2, k, n, user.data(), item.data(), pred.data(), 0) != PCR_OK) {
        fprintf(stderr, "predict: %s\n", pcr_last_error());
        return 1;
    }
    // one "%lf\n" per line (pmf-predict.cpp:63), formatted by up to 16 threads into buffers that go to the file in line order
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n / 65536 + 1));
    const int64_t per_round = (int64_t)1 << 20;
    bool ok = true;
    for (int64_t z0 = 0; z0 < n && ok; z0 += per_round * T) {
        std::vector<std::string> out((size_t)T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                const int64_t a0 = std::min(n, z0 + per_round * t), a1 = std::min(n, a0 + per_round);
                std::string& o = out[(size_t)t];
                o.resize((size_t)(a1 - a0) * 336 + 16);          // (a double in fixed notation: up to 309 integer digits + ".dddddd")
                char* p = &o[0];
                for (int64_t z = a0; z < a1; ++z) { p = std::to_chars(p, p + 334, pred[(size_t)z], std::chars_format::fixed, 6).ptr; *p++ = '\n'; }
                o.resize((size_t)(p - &o[0]));
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < T && ok; ++t) ok = out[(size_t)t].empty() || fwrite(out[(size_t)t].data(), 1, out[(size_t)t].size(), out_fp) == out[(size_t)t].size();
    }
    ok = (fclose(out_fp) == 0) && ok;
    if (!ok) { fprintf(stderr, "short write to %s\n", argv[3]); return 1; }
    // (the output is closed: leave without the HIP runtime's piecewise teardown, as omp-pmf-train does)
    fflush(stdout); fflush(stderr);
    _exit(0);
}
