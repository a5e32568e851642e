// pcr_synth.cpp -- libpcrsynth.so: multi-threaded generator of the shape-matched synthetic rating sets of
// SURVEY.md 8d (bench / test data; NOT part of the drop-in boundary of include/primalcr.h).
//
// Same recipe as primalcr_amd/synth.py (lognormal per-user counts rescaled to the target nnz, uniform item sets without
// replacement, rank-8 ground truth + N(0, 0.5^2) noise cut at the quantiles that reproduce the 1-5 star shares of
// ml1m/test.ratings, n_test held-out ratings per user), but every user draws from its OWN counter-based random stream
// keyed by (seed, user): any user range can be generated on its own, by any number of threads, with the same result --
// 100 M ratings in seconds instead of the minutes the numpy generator takes, and a rank can generate just the users of
// a configs[4] share.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
struct pcr_synth_params {
    int64_t d1, d2, nnz;     // users, items, training ratings of the WHOLE shape
    double mu, sigma;        // lognormal parameters of the per-user counts
    int32_t real_valued;     // 1: rating = score (toy-example-like), 0: 1..5 by quantile thresholds
    int32_t n_test;          // held-out ratings per user
    int32_t min_count;       // lower clip of the per-user training count
    int32_t reserved;
    uint64_t seed;
};
int pcr_synth_counts(const pcr_synth_params* p, int64_t* cnt_train, int64_t* cnt_test);
int pcr_synth_fill(const pcr_synth_params* p, int64_t u0, int64_t u1, const int64_t* cnt_train, const int64_t* cnt_test,
                   int32_t* item, double* val, int32_t* titem, double* tval, int threads);
}

namespace {

inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {                       // xoshiro256**, seeded from (seed, stream, id) through splitmix64
    uint64_t s[4];
    double spare = 0.0;
    bool has_spare = false;
    Rng(uint64_t seed, uint64_t stream, uint64_t id) {
        uint64_t x = seed ^ (stream * 0xD6E8FEB86659FD93ull) ^ (id * 0xA24BAED4963EE407ull + 0x9FB21C651E98DF25ull);
        for (auto& w : s) w = splitmix(x);
    }
    static inline uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    inline uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    inline double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }      // [0, 1)
    inline uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
    inline double normal() {                       // Marsaglia polar
        if (has_spare) { has_spare = false; return spare; }
        double a, b, q;
        do { a = 2.0 * uniform() - 1.0; b = 2.0 * uniform() - 1.0; q = a * a + b * b; } while (q >= 1.0 || q == 0.0);
        const double f = std::sqrt(-2.0 * std::log(q) / q);
        spare = b * f; has_spare = true;
        return a * f;
    }
};

constexpr int GK = 8;                                             // rank of the ground truth
const double LEVEL_SHARES[5] = {3047.0 / 60400, 5541.0 / 60400, 14122.0 / 60400, 21259.0 / 60400, 16431.0 / 60400};

struct Scratch {                     // per thread
    std::vector<uint64_t> bits;      // d2-bit membership map
    std::vector<int32_t> items;
    std::vector<double> score;
    std::vector<uint8_t> is_test;
    std::vector<int32_t> idx;
};

// items (ascending) and scores of user u, who holds `total` ratings (training + held-out)
void gen_user(const pcr_synth_params& P, const std::vector<double>& Vg, int64_t u, int64_t total, Scratch& S) {
    const int64_t d2 = P.d2;
    Rng rng(P.seed, 1, (uint64_t)u);
    S.items.clear();
    if (S.bits.size() != (size_t)((d2 + 63) / 64)) S.bits.assign((d2 + 63) / 64, 0);
    // draw the smaller of the set and its complement by rejection against the bitmap
    const bool complement = total > d2 / 2;
    const int64_t draw = complement ? d2 - total : total;
    std::vector<int32_t>& picked = S.idx;
    picked.clear();
    while ((int64_t)picked.size() < draw) {
        const int64_t j = (int64_t)rng.below((uint64_t)d2);
        uint64_t& w = S.bits[j >> 6];
        const uint64_t m = 1ull << (j & 63);
        if (w & m) continue;
        w |= m;
        picked.push_back((int32_t)j);
    }
    if (!complement && draw * 64 < d2) {                          // sparse: sort the picks, clear their bits
        std::sort(picked.begin(), picked.end());
        for (int32_t j : picked) S.bits[j >> 6] = 0;
        S.items.assign(picked.begin(), picked.end());
    } else {                                                      // dense: walk the bitmap (ascending for free)
        for (int64_t wi = 0; wi < (int64_t)S.bits.size(); ++wi) {
            uint64_t w = complement ? ~S.bits[wi] : S.bits[wi];
            if (wi == (int64_t)S.bits.size() - 1 && (d2 & 63)) w &= (1ull << (d2 & 63)) - 1;
            while (w) {
                const int b = __builtin_ctzll(w);
                S.items.push_back((int32_t)(wi * 64 + b));
                w &= w - 1;
            }
            S.bits[wi] = 0;
        }
    }
    // rank-8 ground truth + noise: score = ug . vg_item + N(0, 0.5^2)
    double ug[GK];
    {
        Rng ru(P.seed, 2, (uint64_t)u);
        for (int t = 0; t < GK; ++t) ug[t] = ru.normal() * std::sqrt(1.0 / GK);
    }
    S.score.resize(S.items.size());
    for (size_t q = 0; q < S.items.size(); ++q) {
        const double* vg = &Vg[(size_t)S.items[q] * GK];
        double s = 0.0;
        for (int t = 0; t < GK; ++t) s += ug[t] * vg[t];
        S.score[q] = s + 0.5 * rng.normal();
    }
}

void item_truth(const pcr_synth_params& P, std::vector<double>& Vg) {
    Vg.resize((size_t)P.d2 * GK);
    for (int64_t j = 0; j < P.d2; ++j) {
        Rng r(P.seed, 3, (uint64_t)j);
        for (int t = 0; t < GK; ++t) Vg[(size_t)j * GK + t] = r.normal();
    }
}

void thresholds(const pcr_synth_params& P, const std::vector<double>& Vg, const int64_t* cnt_train, const int64_t* cnt_test, double thr[4]) {
    // quantiles of the scores of an evenly spread sample of users (a property of the shape, not of the range generated)
    const int64_t S = std::min<int64_t>(P.d1, 4096);
    std::vector<double> all;
    Scratch sc;
    for (int64_t i = 0; i < S; ++i) {
        const int64_t u = i * P.d1 / S;
        gen_user(P, Vg, u, cnt_train[u] + cnt_test[u], sc);
        all.insert(all.end(), sc.score.begin(), sc.score.end());
    }
    std::sort(all.begin(), all.end());
    double cum = 0.0;
    for (int l = 0; l < 4; ++l) {
        cum += LEVEL_SHARES[l];
        const size_t k = all.empty() ? 0 : std::min(all.size() - 1, (size_t)(cum * (double)all.size()));
        thr[l] = all.empty() ? 0.0 : all[k];
    }
}

}  // namespace

extern "C" int pcr_synth_counts(const pcr_synth_params* p, int64_t* cnt_train, int64_t* cnt_test) {
    if (!p || !cnt_train || !cnt_test || p->d1 < 0 || p->d2 < 1 || p->nnz < 0) return -1;
    const int64_t d1 = p->d1, d2 = p->d2;
    const int64_t lo = std::min<int64_t>(p->min_count, d2), hi = std::max<int64_t>(lo, d2 - 20);
    std::vector<double> w(d1);
    double sum = 0.0;
    for (int64_t u = 0; u < d1; ++u) {
        Rng r(p->seed, 0, (uint64_t)u);
        w[u] = std::exp(p->mu + p->sigma * r.normal());
        sum += w[u];
    }
    int64_t tot = 0;
    for (int64_t u = 0; u < d1; ++u) {
        int64_t c = (int64_t)std::llrint(w[u] * ((double)p->nnz / sum));
        c = std::min(hi, std::max(lo, c));
        cnt_train[u] = c; tot += c;
    }
    // fix the total deterministically: one rating at a time over the users that have slack, in index order
    for (int pass = 0; pass < 1024 && tot != p->nnz; ++pass) {
        bool moved = false;
        for (int64_t u = 0; u < d1 && tot != p->nnz; ++u) {
            if (tot < p->nnz && cnt_train[u] < hi) { cnt_train[u]++; tot++; moved = true; }
            else if (tot > p->nnz && cnt_train[u] > lo) { cnt_train[u]--; tot--; moved = true; }
        }
        if (!moved) break;
    }
    for (int64_t u = 0; u < d1; ++u) cnt_test[u] = std::max<int64_t>(0, std::min<int64_t>(p->n_test, d2 - cnt_train[u]));
    return 0;
}

extern "C" int pcr_synth_fill(const pcr_synth_params* p, int64_t u0, int64_t u1, const int64_t* cnt_train, const int64_t* cnt_test,
                              int32_t* item, double* val, int32_t* titem, double* tval, int threads) {
    if (!p || u0 < 0 || u1 < u0 || u1 > p->d1 || !cnt_train || !cnt_test) return -1;
    const pcr_synth_params P = *p;
    std::vector<double> Vg;
    item_truth(P, Vg);
    double thr[4] = {0, 0, 0, 0};
    if (!P.real_valued) thresholds(P, Vg, cnt_train, cnt_test, thr);
    const int64_t nu = u1 - u0;
    std::vector<int64_t> otr(nu + 1, 0), ote(nu + 1, 0);          // output offsets of the range
    for (int64_t i = 0; i < nu; ++i) { otr[i + 1] = otr[i] + cnt_train[u0 + i]; ote[i + 1] = ote[i] + cnt_test[u0 + i]; }
    if (threads <= 0) threads = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    std::atomic<int64_t> next{0};
    const int64_t CH = 256;
    auto work = [&]() {
        Scratch S;
        for (;;) {
            const int64_t a = next.fetch_add(CH);
            if (a >= nu) break;
            for (int64_t i = a; i < std::min(nu, a + CH); ++i) {
                const int64_t u = u0 + i, nt = cnt_test[u], total = cnt_train[u] + nt;
                gen_user(P, Vg, u, total, S);
                // hold out nt ratings: a random subset (partial Fisher-Yates over the positions)
                S.is_test.assign((size_t)total, 0);
                if (nt > 0) {
                    Rng rt(P.seed, 4, (uint64_t)u);
                    S.idx.resize((size_t)total);
                    for (int64_t q = 0; q < total; ++q) S.idx[q] = (int32_t)q;
                    for (int64_t q = 0; q < nt; ++q) {
                        const int64_t j = q + (int64_t)rt.below((uint64_t)(total - q));
                        std::swap(S.idx[q], S.idx[j]);
                        S.is_test[S.idx[q]] = 1;
                    }
                }
                int64_t a_tr = otr[i], a_te = ote[i];
                for (int64_t q = 0; q < total; ++q) {
                    const double s = S.score[q];
                    double v = s;
                    if (!P.real_valued) v = 1.0 + (s >= thr[0]) + (s >= thr[1]) + (s >= thr[2]) + (s >= thr[3]);
                    if (S.is_test[q]) { if (titem) { titem[a_te] = S.items[q]; tval[a_te] = v; } ++a_te; }
                    else { if (item) { item[a_tr] = S.items[q]; val[a_tr] = v; } ++a_tr; }
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return 0;
}

// The reference's rating-file format (util.h:118-131: one "user item rating" per line, 1-based ids) from a CSR, written by
// `threads` formatting threads: every thread formats a contiguous user range into its own buffer, buffers go to the file in
// user order.  Integer-valued ratings are printed as integers ("5"), real-valued ones with 17 significant digits (round trip).
// user_base = 0-based id of the CSR's first user in the file.  Returns 0, or -1 on an I/O error.
#include <cstdio>
#include <string>
extern "C" int pcr_synth_write_text(const char* path, int64_t nu, int64_t user_base, const int64_t* index, const int32_t* item,
                                    const double* val, int threads) {
    if (!path || nu < 0 || !index || (index[nu] > 0 && (!item || !val))) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    if (threads <= 0) threads = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    const int64_t nnz = index[nu];
    const int64_t per_round = (int64_t)4 << 20;                  // ratings formatted per round and thread
    bool ok = true;
    auto put_uint = [](char* p, uint64_t x) {
        char tmp[24]; int n = 0;
        do { tmp[n++] = (char)('0' + x % 10); x /= 10; } while (x);
        for (int i = 0; i < n; ++i) p[i] = tmp[n - 1 - i];
        return p + n;
    };
    for (int64_t u_lo = 0; u_lo < nu && ok;) {
        // this round's user ranges: about per_round ratings per thread
        std::vector<int64_t> cut(1, u_lo);
        for (int t = 0; t < threads && cut.back() < nu; ++t) {
            const int64_t target = index[cut.back()] + per_round;
            int64_t u = (int64_t)(std::upper_bound(index + cut.back(), index + nu + 1, target) - index);
            cut.push_back(std::min(nu, std::max(cut.back() + 1, u - 1)));
        }
        const int parts = (int)cut.size() - 1;
        std::vector<std::string> out((size_t)parts);
        std::vector<std::thread> pool;
        for (int t = 0; t < parts; ++t)
            pool.emplace_back([&, t]() {
                std::string& s = out[(size_t)t];
                s.resize((size_t)(index[cut[t + 1]] - index[cut[t]]) * 48 + 64);
                char* p = &s[0];
                for (int64_t u = cut[t]; u < cut[t + 1]; ++u)
                    for (int64_t z = index[u]; z < index[u + 1]; ++z) {
                        p = put_uint(p, (uint64_t)(user_base + u + 1)); *p++ = ' ';
                        p = put_uint(p, (uint64_t)item[z] + 1); *p++ = ' ';
                        const double v = val[z];
                        if (v == std::floor(v) && std::fabs(v) < 1e15) {
                            if (v < 0) *p++ = '-';
                            p = put_uint(p, (uint64_t)std::fabs(v));
                        } else {
                            p += snprintf(p, 32, "%.17g", v);
                        }
                        *p++ = '\n';
                    }
                s.resize((size_t)(p - &s[0]));
            });
        for (auto& th : pool) th.join();
        for (int t = 0; t < parts && ok; ++t) ok = out[(size_t)t].empty() || fwrite(out[(size_t)t].data(), 1, out[(size_t)t].size(), f) == out[(size_t)t].size();
        u_lo = cut.back();
    }
    (void)nnz;
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}
