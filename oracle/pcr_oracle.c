/*
 * pcr_oracle.c -- CPU ORACLE (test infrastructure only; see pcr_oracle.h).
 *
 * Plain-C fp64 restatement of wuliwei9278/primalCR's PrimalCR++ / PrimalCR
 * training path.  Loop order, comparison operators and summation order follow
 * the reference so the numbers agree with the compiled reference to rounding.
 * All reference citations are relative to /root/reference/.
 */
#include "pcr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* small helpers                                                      */
/* ------------------------------------------------------------------ */

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* util.cpp:103-109 dot(); the reference iterates i = n-1 .. 0 */
static double dotv(const double *a, const double *b, long n) {
    double ret = 0;
    for (long i = n - 1; i >= 0; --i) ret += a[i] * b[i];
    return ret;
}
/* util.cpp:126-132 norm(): the SQUARED 2-norm */
static double norm2(const double *a, long n) {
    double ret = 0;
    for (long i = n - 1; i >= 0; --i) ret += a[i] * a[i];
    return ret;
}
/* util.cpp:133-138 norm(mat_t): rows visited last-to-first */
static double norm2_mat(const double *M, long rows, int r) {
    double reg = 0;
    for (long i = rows - 1; i >= 0; --i) reg += norm2(M + i * r, r);
    return reg;
}

/* stable merge sort of idx[0..n) by key[idx] ascending (the reference uses
 * std::sort, whose order among equal keys is unspecified and irrelevant to
 * the sums computed from it) */
static void msort_idx(long *idx, long *tmp, const double *key, long n) {
    if (n < 2) return;
    long h = n / 2;
    msort_idx(idx, tmp, key, h);
    msort_idx(idx + h, tmp, key, n - h);
    long a = 0, b = h, o = 0;
    while (a < h && b < n) tmp[o++] = (key[idx[b]] < key[idx[a]]) ? idx[b++] : idx[a++];
    while (a < h) tmp[o++] = idx[a++];
    while (b < n) tmp[o++] = idx[b++];
    memcpy(idx, tmp, (size_t)n * sizeof(long));
}

static int cmp_long(const void *a, const void *b) {
    long x = *(const long *)a, y = *(const long *)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------ */
/* util.cpp:80-93 initial()                                            */
/* ------------------------------------------------------------------ */

typedef struct {
    unsigned long x;      /* minstd_rand0 state, default seed 1 */
    int saved_ok;
    double saved;
} orc_rng_t;

static unsigned long minstd0(orc_rng_t *g) {
    g->x = (g->x * 16807UL) % 2147483647UL;
    return g->x;
}
/* libstdc++ generate_canonical<double,53>(minstd_rand0): range 2147483646,
 * floor(log2) = 30 -> k = 2 draws */
static double canonical(orc_rng_t *g) {
    const long double R = 2147483646.0L;
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) {
        sum += (double)(minstd0(g) - 1UL) * tmp;
        tmp = (double)((long double)tmp * R);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}
/* libstdc++ normal_distribution<double>::operator(): Marsaglia polar,
 * returns y*mult first and keeps x*mult for the next call */
static double normal01(orc_rng_t *g) {
    if (g->saved_ok) {
        g->saved_ok = 0;
        return g->saved;
    }
    double x, y, r2;
    do {
        x = 2.0 * canonical(g) - 1.0;
        y = 2.0 * canonical(g) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = sqrt(-2 * log(r2) / r2);
    g->saved = x * mult;
    g->saved_ok = 1;
    return y * mult;
}

void orc_initial(double *X, long n, long k) {
    orc_rng_t g = {1UL, 0, 0.0};
    for (long i = 0; i < n; ++i)
        for (long j = 0; j < k; ++j) X[i * k + j] = normal01(&g);
}

/* ------------------------------------------------------------------ */
/* CSR construction                                                   */
/* ------------------------------------------------------------------ */

typedef struct { int u, i; long pos; } trip_t;
static int cmp_trip(const void *a, const void *b) {
    const trip_t *x = (const trip_t *)a, *y = (const trip_t *)b;
    if (x->u != y->u) return (x->u > y->u) - (x->u < y->u);
    if (x->i != y->i) return (x->i > y->i) - (x->i < y->i);
    return (x->pos > y->pos) - (x->pos < y->pos);
}

/* util.h:223-247: entries sorted by (row=user, col=item); util.cpp:229-243
 * walks them user by user */
int orc_build_csr(long d1, long nnz, const int *tu, const int *ti,
                  const double *tv, long *idx, long *item, double *val) {
    trip_t *t = (trip_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(trip_t));
    if (!t) return -1;
    for (long z = 0; z < nnz; ++z) { t[z].u = tu[z]; t[z].i = ti[z]; t[z].pos = z; }
    qsort(t, (size_t)nnz, sizeof(trip_t), cmp_trip);
    long cc = 0;
    for (long u = 0; u < d1; ++u) {
        idx[u] = cc;
        while (cc < nnz && t[cc].u == u) {
            item[cc] = t[cc].i;
            val[cc] = tv[t[cc].pos];
            ++cc;
        }
    }
    idx[d1] = nnz;
    free(t);
    return 0;
}

/* util.cpp:250-274 */
long orc_build_csr_test(long d1, long nnz, const int *tu, const int *ti,
                        const double *tv, long *idx, long *item, double *val) {
    long cc = 0;
    for (long j = 0; j < d1; ++j) {
        idx[j] = cc;
        for (; cc < nnz; ++cc) {
            if (tu[cc] > j) break;
            val[cc] = tv[cc];
            item[cc] = ti[cc];
        }
    }
    idx[d1] = cc;
    return cc;
}

/* ------------------------------------------------------------------ */
/* per-user sorted bundle (pcrpp.cpp:38-137, 447-477 infor_ui)        */
/* ------------------------------------------------------------------ */

typedef struct {
    long len, num_levels;
    double *mm_sorted;   /* len */
    long *lvl_sorted;    /* len, dense level index 0..T-1 */
    long *item_sorted;   /* len (d2bar_sorted) */
    long *perm;          /* len: sorted position -> offset in the user segment */
    long *count_right;   /* T (get_count_right) */
    /* scratch */
    long *tmp, *levels;
    long cap, lcap;
} ubundle_t;

static void ub_init(ubundle_t *b) { memset(b, 0, sizeof(*b)); }
static void ub_free(ubundle_t *b) {
    free(b->mm_sorted); free(b->lvl_sorted); free(b->item_sorted);
    free(b->perm); free(b->count_right); free(b->tmp); free(b->levels);
    ub_init(b);
}
static void ub_reserve(ubundle_t *b, long len) {
    if (len <= b->cap) return;
    long c = len + 16;
    b->mm_sorted = (double *)realloc(b->mm_sorted, (size_t)c * sizeof(double));
    b->lvl_sorted = (long *)realloc(b->lvl_sorted, (size_t)c * sizeof(long));
    b->item_sorted = (long *)realloc(b->item_sorted, (size_t)c * sizeof(long));
    b->perm = (long *)realloc(b->perm, (size_t)c * sizeof(long));
    b->tmp = (long *)realloc(b->tmp, (size_t)c * sizeof(long));
    b->levels = (long *)realloc(b->levels, (size_t)c * sizeof(long));
    b->count_right = (long *)realloc(b->count_right, (size_t)c * sizeof(long));
    b->cap = c;
}

/* mm: the user's m segment (len values, in CSR order); val/item: the user's
 * CSR segment.  Follows find_levels (:38-49), get_sorted_mm (:52-83),
 * get_sorted_vals (:87-95), get_sorted_d2bar (:99-107), the level remap
 * (:182-189) and get_count_right (:129-137). */
static void ub_build(ubundle_t *b, const double *mm, const double *val,
                     const long *item, long len) {
    ub_reserve(b, len);
    b->len = len;
    /* find_levels: sorted distinct lround(val) */
    long T = 0;
    for (long j = 0; j < len; ++j) b->levels[j] = lround(val[j]);
    if (len > 1) qsort(b->levels, (size_t)len, sizeof(long), cmp_long);     /* (an empty user has no arrays: UBSan, nonnull) */
    for (long j = 0; j < len; ++j)
        if (j == 0 || b->levels[j] != b->levels[T - 1]) b->levels[T++] = b->levels[j];
    b->num_levels = T;
    for (long j = 0; j < len; ++j) b->perm[j] = j;
    msort_idx(b->perm, b->tmp, mm, len);
    for (long k = 0; k < T; ++k) b->count_right[k] = 0;
    for (long j = 0; j < len; ++j) {
        long p = b->perm[j];
        b->mm_sorted[j] = mm[p];
        b->item_sorted[j] = item ? item[p] : 0;
        long lv = lround(val[p]);
        long k = 0;
        while (b->levels[k] != lv) ++k;
        b->lvl_sorted[j] = k;
        b->count_right[k] += 1;
    }
}

/* The two-pointer sweep shared by obtain_g_new (:190-238), compute_Ha_new
 * (:287-318), obtain_g_u_new (:506-535) and obtain_Hs_new (:595-621).
 * x = mm_sorted for the gradient (shift = 1: terms (x_j -/+ 1)), x = b_sorted
 * for Hessian-vector products (shift = 0).  Writes c[j] (already *2). */
static void sweep_c(const ubundle_t *b, const double *x, double shift, double *c) {
    long len = b->len, T = b->num_levels;
    const double *mm = b->mm_sorted;
    const long *lv = b->lvl_sorted;
    double *right_sum = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
    double *left_sum = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
    long *cnt_left = (long *)calloc((size_t)(T > 0 ? T : 1), sizeof(long));
    long *cnt_right = (long *)calloc((size_t)(T > 0 ? T : 1), sizeof(long));
    for (long j = 0; j < len; ++j) right_sum[lv[j]] += x[j];      /* get_levels_sum */
    for (long k = 0; k < T; ++k) cnt_right[k] = b->count_right[k];
    long now_left = 0, now_right = 0;
    for (long j = 0; j < len; ++j) {
        double now_cut = mm[j];
        long now_val = lv[j];
        while (now_left < len && mm[now_left] <= now_cut + 1.0) {
            long level = lv[now_left];
            left_sum[level] += x[now_left];
            cnt_left[level] += 1;
            now_left += 1;
        }
        while (now_right < len && mm[now_right] < now_cut - 1.0) {
            long level = lv[now_right];
            right_sum[level] -= x[now_right];
            cnt_right[level] -= 1;
            now_right += 1;
        }
        double cc = 0.0;
        for (long k = 0; k <= now_val - 1; ++k)
            cc += (cnt_right[k] * (x[j] - shift) - right_sum[k]);
        for (long k = now_val + 1; k < T; ++k)
            cc += (cnt_left[k] * (x[j] + shift) - left_sum[k]);
        c[j] = cc * 2.0;
    }
    free(right_sum); free(left_sum); free(cnt_left); free(cnt_right);
}

/* objective sweep: pcrpp.cpp:388-407 and :552-571 */
static double sweep_obj(const ubundle_t *b) {
    long len = b->len, T = b->num_levels;
    const double *mm = b->mm_sorted;
    const long *lv = b->lvl_sorted;
    double *left_sum = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
    double *left_sq = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
    long *cnt_left = (long *)calloc((size_t)(T > 0 ? T : 1), sizeof(long));
    long now_left = 0;
    double res = 0.0;
    for (long j = 0; j < len; ++j) {
        double now_cut = mm[j];
        long now_val = lv[j];
        while (now_left < len && mm[now_left] <= now_cut + 1.0) {
            long level = lv[now_left];
            left_sum[level] += (mm[now_left] - 1.0);
            left_sq[level] += pow(mm[now_left] - 1.0, 2.0);
            cnt_left[level] += 1;
            now_left += 1;
        }
        for (long k = now_val + 1; k < T; ++k)
            res += (cnt_left[k] * pow(now_cut, 2.0) - 2.0 * now_cut * left_sum[k] + left_sq[k]);
    }
    free(left_sum); free(left_sq); free(cnt_left);
    return res;
}

/* ------------------------------------------------------------------ */
/* PrimalCR++ V side                                                  */
/* ------------------------------------------------------------------ */

void orc_comp_m(const double *U, const double *V, long d1, const long *idx,
                const long *item, int r, double *m) {
    for (long u = 0; u < d1; ++u)
        for (long z = idx[u]; z < idx[u + 1]; ++z) {
            const double *uu = U + u * r, *vv = V + item[z] * r;
            double dot_res = 0;
            for (int j = 0; j < r; ++j) dot_res = dot_res + uu[j] * vv[j];
            m[z] = dot_res;
        }
}

double orc_objective_new(const double *m, const double *U, const double *V,
                         long d1, long d2, const long *idx, const double *val,
                         int r, double lambda) {
    double res = 0.0;
    double norm_U = norm2_mat(U, d1, r);
    double norm_V = norm2_mat(V, d2, r);
    ubundle_t b; ub_init(&b);
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], len = idx[i + 1] - idx[i];
        ub_build(&b, m + start, val + start, NULL, len);
        res += sweep_obj(&b);
    }
    ub_free(&b);
    res += lambda * (norm_U + norm_V) / 2.0;
    return res;
}

void orc_obtain_g_new(const double *U, const double *V, long d1, long d2,
                      const long *idx, const long *item, const double *val,
                      const double *m, int r, double lambda, double *g) {
    for (long z = 0; z < d2 * r; ++z) g[z] = V[z] * lambda;     /* copy_mat_t(V, lambda) */
    ubundle_t b; ub_init(&b);
    double *c = NULL; long ccap = 0;
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], len = idx[i + 1] - idx[i];
        if (len > ccap) { ccap = len + 16; c = (double *)realloc(c, (size_t)ccap * sizeof(double)); }
        ub_build(&b, m + start, val + start, item + start, len);
        sweep_c(&b, b.mm_sorted, 1.0, c);
        for (long j = 0; j < len; ++j) {
            long p = b.item_sorted[j];
            for (long k = 0; k < r; ++k) g[p * r + k] += c[j] * U[i * r + k];
        }
    }
    free(c);
    ub_free(&b);
}

void orc_compute_Ha_new(const double *a, const double *m, const double *U,
                        long d1, long d2, const long *idx, const long *item,
                        const double *val, int r, double lambda, double *Ha) {
    for (long z = 0; z < d2 * r; ++z) Ha[z] = a[z] * lambda;    /* copy_vec_t(a, lambda) */
    ubundle_t b; ub_init(&b);
    double *c = NULL, *bb = NULL, *bs = NULL; long ccap = 0;
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], len = idx[i + 1] - idx[i];
        if (len > ccap) {
            ccap = len + 16;
            c = (double *)realloc(c, (size_t)ccap * sizeof(double));
            bb = (double *)realloc(bb, (size_t)ccap * sizeof(double));
            bs = (double *)realloc(bs, (size_t)ccap * sizeof(double));
        }
        for (long k = 0; k < len; ++k) {                         /* :266-271 vec_prod_array */
            long q = item[start + k];
            double res = 0.0;
            for (long t = 0; t < r; ++t) res += U[i * r + t] * a[q * r + t];
            bb[k] = res;
        }
        ub_build(&b, m + start, val + start, item + start, len);
        for (long j = 0; j < len; ++j) bs[j] = bb[b.perm[j]];    /* get_sorted_b */
        sweep_c(&b, bs, 0.0, c);
        for (long j = 0; j < len; ++j) {
            long p = b.item_sorted[j];
            for (long ii = 0; ii < r; ++ii) {
                double tmp = c[j] * U[i * r + ii];
                Ha[p * r + ii] += tmp;
            }
        }
    }
    free(c); free(bb); free(bs);
    ub_free(&b);
}

/* The CG recurrence of solve_delta_new (:335-358) / solve_delta_u_new
 * (:628-647) / pcr.cpp:248-277, :498-520, parameterised by the Hv callback. */
typedef void (*hv_fn)(const double *p, double *Hp, void *ctx);

/* The reference hard-codes 10 iterations and 1 % (:340,344); tests of the product's cg_max_iter /
 * cg_tol extension (include/primalcr.h) move them with orc_set_cg. */
static int g_cg_max = 10;
static double g_cg_tol = 0.01;
void orc_set_cg(int max_iter, double tol) { g_cg_max = max_iter; g_cg_tol = tol; }

static int cg_solve(const double *g, long n, hv_fn hv, void *ctx, double *delta) {
    double *rr = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *Hp = (double *)malloc((size_t)n * sizeof(double));
    for (long i = 0; i < n; ++i) { delta[i] = 0.0; rr[i] = g[i] * -1.0; p[i] = g[i]; }
    double err = sqrt(norm2(rr, n)) * g_cg_tol;
    int its = 0;
    for (int k = 1; k <= g_cg_max; ++k) {
        hv(p, Hp, ctx);
        ++its;
        double prod_p_Hp = dotv(p, Hp, n);
        double alpha = -1.0 * dotv(rr, p, n) / prod_p_Hp;
        for (long i = 0; i < n; ++i) delta[i] = delta[i] * 1.0 + p[i] * alpha;
        for (long i = 0; i < n; ++i) rr[i] = rr[i] * 1.0 + Hp[i] * alpha;
        if (sqrt(norm2(rr, n)) < err) break;
        double bq = dotv(rr, Hp, n) / prod_p_Hp;
        for (long i = 0; i < n; ++i) p[i] = rr[i] * -1.0 + p[i] * bq;
    }
    free(rr); free(p); free(Hp);
    return its;
}

typedef struct {
    const double *m, *U, *val; const long *idx, *item;
    long d1, d2; int r; double lambda; int solver;
} vctx_t;

static void hv_V(const double *p, double *Hp, void *ctx) {
    vctx_t *c = (vctx_t *)ctx;
    if (c->solver == 2)
        orc_compute_Ha_new(p, c->m, c->U, c->d1, c->d2, c->idx, c->item, c->val, c->r, c->lambda, Hp);
    else
        orc_compute_Ha(p, c->m, c->U, c->d1, c->d2, c->idx, c->item, c->val, c->r, c->lambda, Hp);
}

int orc_solve_delta_new(const double *g, const double *m, const double *U,
                        long d1, long d2, const long *idx, const long *item,
                        const double *val, int r, double lambda, double *delta) {
    vctx_t c = {m, U, val, idx, item, d1, d2, r, lambda, 2};
    return cg_solve(g, d2 * (long)r, hv_V, &c, delta);
}

/* shared by update_V_new (pcrpp.cpp:415-444) and update_V (pcr.cpp:279-330) */
static int update_V_any(int solver, long d1, long d2, const long *idx,
                        const long *item, const double *val, double lambda,
                        double stepsize, int r, const double *U, double *V,
                        double *now_obj, double *m, int *accepted, int *cg_iters) {
    long n = d2 * (long)r;
    orc_comp_m(U, V, d1, idx, item, r, m);
    double *g = (double *)malloc((size_t)n * sizeof(double));
    double *delta = (double *)malloc((size_t)n * sizeof(double));
    double *V_new = (double *)malloc((size_t)n * sizeof(double));
    if (solver == 2) orc_obtain_g_new(U, V, d1, d2, idx, item, val, m, r, lambda, g);
    else orc_obtain_g(U, V, d1, d2, idx, item, val, m, r, lambda, g);
    vctx_t c = {m, U, val, idx, item, d1, d2, r, lambda, solver};
    int its = cg_solve(g, n, hv_V, &c, delta);
    if (cg_iters) *cg_iters = its;
    double prev_obj = (solver == 2)
        ? orc_objective_new(m, U, V, d1, d2, idx, val, r, lambda)
        : orc_objective(m, U, V, d1, d2, idx, val, r, lambda);
    int tries = 0, acc = 0;
    for (int iter = 0; iter < 20; ++iter) {
        for (long i = 0; i < n; ++i) { V_new[i] = V[i]; V_new[i] -= stepsize * delta[i]; }
        orc_comp_m(U, V_new, d1, idx, item, r, m);
        *now_obj = (solver == 2)
            ? orc_objective_new(m, U, V_new, d1, d2, idx, val, r, lambda)
            : orc_objective(m, U, V_new, d1, d2, idx, val, r, lambda);
        ++tries;
        if (*now_obj < prev_obj) {
            memcpy(V, V_new, (size_t)n * sizeof(double));
            acc = 1;
            break;
        } else {
            stepsize /= 2.0;
        }
    }
    if (accepted) *accepted = acc;
    free(g); free(delta); free(V_new);
    return tries;
}

int orc_update_V_new(long d1, long d2, const long *idx, const long *item,
                     const double *val, double lambda, double stepsize, int r,
                     const double *U, double *V, double *now_obj,
                     double *m_out, int *accepted, int *cg_iters) {
    return update_V_any(2, d1, d2, idx, item, val, lambda, stepsize, r, U, V,
                        now_obj, m_out, accepted, cg_iters);
}

/* ------------------------------------------------------------------ */
/* PrimalCR++ U side                                                  */
/* ------------------------------------------------------------------ */

typedef struct { const ubundle_t *b; const double *V; int r; double lambda; double *c, *bs; } uctx_t;

/* obtain_Hs_new, pcrpp.cpp:576-625 */
static void hv_u(const double *s, double *Hs, void *ctx) {
    uctx_t *u = (uctx_t *)ctx;
    const ubundle_t *b = u->b;
    int r = u->r;
    for (int t = 0; t < r; ++t) Hs[t] = s[t] * u->lambda;
    for (long k = 0; k < b->len; ++k)
        u->bs[k] = dotv(s, u->V + b->item_sorted[k] * r, r);
    sweep_c(b, u->bs, 0.0, u->c);
    for (long j = 0; j < b->len; ++j) {
        const double *vp = u->V + b->item_sorted[j] * r;
        for (int t = 0; t < r; ++t) Hs[t] = Hs[t] * 1.0 + vp[t] * u->c[j];
    }
}

int orc_update_u_new(long i, const double *V, const long *idx,
                     const long *item, const double *val, const double *m,
                     int r, double lambda, double stepsize, const double *ui,
                     double *ui_new, double *obj_u_new, int *n_ls) {
    long start = idx[i], len = idx[i + 1] - idx[i];
    ubundle_t b, b2; ub_init(&b); ub_init(&b2);
    ub_build(&b, m + start, val + start, item + start, len);      /* precompute_ui :447-477 */
    double *g = (double *)malloc((size_t)r * sizeof(double));
    double *delta = (double *)calloc((size_t)r, sizeof(double));
    double *c = (double *)malloc((size_t)(len + 1) * sizeof(double));
    double *bs = (double *)malloc((size_t)(len + 1) * sizeof(double));
    double *mm = (double *)malloc((size_t)(len + 1) * sizeof(double));
    /* obtain_g_u_new :493-539 */
    if (len == 0) {
        for (int t = 0; t < r; ++t) g[t] = 0.0;
    } else {
        for (int t = 0; t < r; ++t) g[t] = ui[t] * lambda;
        sweep_c(&b, b.mm_sorted, 1.0, c);
        for (long j = 0; j < len; ++j) {
            const double *vp = V + b.item_sorted[j] * r;
            for (int t = 0; t < r; ++t) g[t] = g[t] * 1.0 + vp[t] * c[j];
        }
    }
    /* objective_u_new :542-573 */
    double prev_obj = 0.0;
    prev_obj += lambda / 2.0 * norm2(ui, r);
    prev_obj += sweep_obj(&b);
    int its = 0, tries = 0;
    if (norm2(g, r) < 0.0001) {                                    /* :787-790 */
        *obj_u_new = prev_obj;
        memcpy(ui_new, ui, (size_t)r * sizeof(double));
    } else {
        uctx_t u = {&b, V, r, lambda, c, bs};
        /* solve_delta_u_new :628-647: rr = -g, p = -rr = g */
        its = cg_solve(g, r, hv_u, &u, delta);
        memcpy(ui_new, ui, (size_t)r * sizeof(double));
        for (int iter = 0; iter < 20; ++iter) {                     /* :794-813 */
            for (int t = 0; t < r; ++t) ui_new[t] = ui[t] * 1.0 + delta[t] * -stepsize;
            for (long j = 0; j < len; ++j) {                        /* compute_mm_old :728-744 */
                const double *vp = V + item[start + j] * r;
                double res = 0.0;
                for (int k = 0; k < r; ++k) res += ui_new[k] * vp[k];
                mm[j] = res;
            }
            ub_build(&b2, mm, val + start, item + start, len);      /* update_infor_ui :684-726 */
            double o = 0.0;
            o += lambda / 2.0 * norm2(ui_new, r);
            o += sweep_obj(&b2);
            *obj_u_new = o;
            ++tries;
            if (*obj_u_new < prev_obj) break;
            else stepsize /= 2.0;
        }
    }
    if (n_ls) *n_ls = tries;
    free(g); free(delta); free(c); free(bs); free(mm);
    ub_free(&b); ub_free(&b2);
    return its;
}

void orc_update_U_new(long d1, long d2, const long *idx, const long *item,
                      const double *val, const double *m, double lambda,
                      double stepsize, int r, const double *V,
                      const double *U, double *U_new, double *now_obj,
                      long *total_cg, long *total_ls) {
    double total = 0.0;
    long tcg = 0, tls = 0;
    for (long i = 0; i < d1; ++i) {
        double obj_u = 0.0; int nls = 0;
        tcg += orc_update_u_new(i, V, idx, item, val, m, r, lambda, stepsize,
                                U + i * r, U_new + i * r, &obj_u, &nls);
        tls += nls;
        total += obj_u;
    }
    total += lambda / 2.0 * norm2_mat(V, d2, r);
    *now_obj = total;
    if (total_cg) *total_cg = tcg;
    if (total_ls) *total_ls = tls;
}

/* ------------------------------------------------------------------ */
/* evaluator util.cpp:434-542                                         */
/* ------------------------------------------------------------------ */

typedef struct { double key; long id; } kv_t;
/* descending by key; ties: the reference's std::sort leaves tie order
 * unspecified -- this oracle keeps the lower original index first */
static int cmp_kv_desc(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key > y->key) return -1;
    if (x->key < y->key) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

void orc_eval(const double *U, const double *V, long d1, const long *idx,
              const long *item, const double *val, int r, int ndcg_k,
              double *pairwise_err, double *ndcg) {
    double sum_error = 0.0, ndcg_sum = 0.0;
    long total_count = 0, total_pair_d1 = 0;
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], end = idx[i + 1] - 1, len = end - start + 1;
        if (len == 0) continue;
        total_count += 1;
        double *score = (double *)malloc((size_t)len * sizeof(double));
        for (long k = 0; k < len; ++k) score[k] = dotv(U + i * r, V + item[start + k] * r, r);
        long error_comps_i = 0, num_comps_i = 0;
        for (long j = start; j < end; ++j) {
            double val_j = val[j];
            for (long k = j + 1; k <= end; ++k) {
                double val_k = val[k];
                if (score[j - start] >= score[k - start] && val_j < val_k) error_comps_i++;
                if (score[j - start] <= score[k - start] && val_j > val_k) error_comps_i++;
                num_comps_i++;
            }
        }
        if (num_comps_i != 0) {
            sum_error += (double)error_comps_i / (double)num_comps_i;
            total_pair_d1++;
        }
        kv_t *bs = (kv_t *)malloc((size_t)len * sizeof(kv_t));
        kv_t *bv = (kv_t *)malloc((size_t)len * sizeof(kv_t));
        for (long k = 0; k < len; ++k) { bs[k].key = score[k]; bs[k].id = k; bv[k].key = val[start + k]; bv[k].id = k; }
        qsort(bs, (size_t)len, sizeof(kv_t), cmp_kv_desc);
        qsort(bv, (size_t)len, sizeof(kv_t), cmp_kv_desc);
        double dcg = 0.0, dcg_max = 0.0;
        long nowk = ndcg_k;
        if (len < nowk) nowk = len;
        for (long k = 1; k <= nowk; ++k) {
            dcg += (pow(2.0, val[start + bs[k - 1].id]) - 1.0) / log2((double)k + 1.0);
            dcg_max += (pow(2.0, val[start + bv[k - 1].id]) - 1.0) / log2((double)k + 1.0);
        }
        ndcg_sum += dcg / dcg_max;
        free(score); free(bs); free(bv);
    }
    *pairwise_err = sum_error / (double)total_pair_d1;
    *ndcg = ndcg_sum / (double)total_count;
}

/* ------------------------------------------------------------------ */
/* PrimalCR (solver 1)                                                 */
/* ------------------------------------------------------------------ */

double orc_objective(const double *m, const double *U, const double *V,
                     long d1, long d2, const long *idx, const double *val,
                     int r, double lambda) {
    double res = 0;
    double norm_U = norm2_mat(U, d1, r), norm_V = norm2_mat(V, d2, r);
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], end = idx[i + 1] - 1;
        for (long j = start; j <= end - 1; ++j) {
            double val_j = val[j];
            for (long k = j + 1; k <= end; ++k) {
                double val_k = val[k];
                if (val_j == val_k) continue;
                double mask = m[j] - m[k];
                if (val_j < val_k) mask = -mask;
                if (mask < 1.0) res += (1.0 - mask) * (1.0 - mask);
            }
        }
    }
    res += lambda * (norm_U + norm_V) / 2.0;
    return res;
}

void orc_obtain_g(const double *U, const double *V, long d1, long d2,
                  const long *idx, const long *item, const double *val,
                  const double *m, int r, double lambda, double *g) {
    for (long z = 0; z < d2 * r; ++z) g[z] = V[z] * lambda;
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], end = idx[i + 1] - 1, len = end - start + 1;
        double *t = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
        for (long j = start; j <= end - 1; ++j) {
            double val_j = val[j];
            for (long k = j + 1; k <= end; ++k) {
                double val_k = val[k];
                double y_ijk = 1.0;
                if (val_j == val_k) continue;
                else if (val_j < val_k) y_ijk = -1.0;
                double mask = m[j] - m[k];
                mask *= y_ijk;
                if (mask < 1.0) {
                    double s_jk = 2.0 * (mask - 1);
                    t[j - start] += s_jk * y_ijk;
                    t[k - start] -= s_jk * y_ijk;
                }
            }
        }
        for (long k = 0; k < len; ++k) {
            long j = item[start + k];
            double c = t[k];
            for (int q = 0; q < r; ++q) g[j * r + q] += c * U[i * r + q];
        }
        free(t);
    }
}

void orc_compute_Ha(const double *a, const double *m, const double *U,
                    long d1, long d2, const long *idx, const long *item,
                    const double *val, int r, double lambda, double *Ha) {
    for (long z = 0; z < d2 * r; ++z) Ha[z] = a[z] * lambda;
    for (long i = 0; i < d1; ++i) {
        long start = idx[i], end = idx[i + 1] - 1, len = end - start + 1;
        double *b = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
        double *cp = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
        for (long k = 0; k < len; ++k) {
            long q = item[start + k];
            double res = 0.0;
            for (long t = 0; t < r; ++t) res += U[i * r + t] * a[q * r + t];
            b[k] = res;
        }
        for (long j = start; j < end; ++j) {
            double val_j = val[j];
            for (long k = j + 1; k <= end; ++k) {
                double val_k = val[k];
                if (val_j == val_k) continue;
                double mask = m[j] - m[k];
                if (val_k > val_j) mask = -mask;
                if (mask < 1.0) {
                    double ddd = b[j - start] - b[k - start];
                    ddd *= 2;
                    cp[j - start] += ddd;
                    cp[k - start] -= ddd;
                }
            }
        }
        for (long k = 0; k < len; ++k) {
            long p = item[start + k];
            double c = cp[k];
            for (long j = 0; j < r; ++j) Ha[p * r + j] += c * U[i * r + j];
        }
        free(b); free(cp);
    }
}

/* objective_u, pcr.cpp:396-427 */
static double objective_u_pairs(const double *mm, const double *ui, const double *val,
                                long len, int r, double lambda) {
    double res = 0.0;
    res += lambda / 2.0 * norm2(ui, r);
    for (long j = 0; j + 1 < len; ++j) {
        double val_j = val[j];
        for (long k = j + 1; k < len; ++k) {
            double val_k = val[k];
            if (val_j == val_k) continue;
            double mask = mm[j] - mm[k];
            if (val_j < val_k) mask = -mask;
            if (mask < 1.0) res += (1.0 - mask) * (1.0 - mask);
        }
    }
    return res;
}

typedef struct {
    const double *V, *val, *D; const long *item; long len; int r; double lambda;
} u1ctx_t;

/* obtain_Hs, pcr.cpp:430-496: active-pair mask D frozen at the gradient point */
static void hv_u1(const double *s, double *Hs, void *ctx) {
    u1ctx_t *u = (u1ctx_t *)ctx;
    long len = u->len; int r = u->r;
    for (int t = 0; t < r; ++t) Hs[t] = s[t] * u->lambda;
    double *b = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
    double *cp = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
    for (long k = 0; k < len; ++k) {
        const double *vp = u->V + u->item[k] * r;
        double res = 0.0;
        for (int q = 0; q < r; ++q) res += s[q] * vp[q];
        b[k] = res;
    }
    long cc = 0;
    for (long j = 0; j + 1 < len; ++j)
        for (long k = j + 1; k < len; ++k) {
            if (u->val[j] == u->val[k]) continue;
            if (u->D[cc] > 0.0) {
                double ddd = b[j] - b[k];
                ddd *= 2.0;
                cp[j] += ddd;
                cp[k] -= ddd;
            }
            cc++;
        }
    for (long k = 0; k < len; ++k) {
        const double *vp = u->V + u->item[k] * r;
        for (int t = 0; t < r; ++t) Hs[t] = Hs[t] * 1.0 + vp[t] * cp[k];
    }
    free(b); free(cp);
}

int orc_update_u(long i, const double *V, const long *idx, const long *item,
                 const double *val, const double *m, int r, double lambda,
                 double stepsize, const double *ui, double *ui_new,
                 double *obj_u_new, int *n_ls) {
    long start = idx[i], len = idx[i + 1] - idx[i];
    size_t num_pairs = (size_t)(len * (len - 1) / 2);
    double *D = (double *)malloc((num_pairs > 0 ? num_pairs : 1) * sizeof(double));
    for (size_t q = 0; q < num_pairs; ++q) D[q] = -1.0;
    double *g = (double *)malloc((size_t)r * sizeof(double));
    double *delta = (double *)calloc((size_t)r, sizeof(double));
    double *t = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
    double *mm = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
    for (int q = 0; q < r; ++q) g[q] = ui[q] * lambda;
    long cc = 0;
    /* obtain_g_u, pcr.cpp:332-394 */
    for (long j = 0; j + 1 < len; ++j) {
        double val_j = val[start + j];
        for (long k = j + 1; k < len; ++k) {
            double val_k = val[start + k];
            if (val_j == val_k) continue;
            double mask = m[start + j] - m[start + k];
            if (val_k > val_j) mask = -mask;
            if (mask < 1.0) {
                D[cc] = 1.0;
                double s_jk = 2 * (1 - mask);
                if (val_k > val_j) s_jk = -s_jk;
                t[j] -= s_jk;
                t[k] += s_jk;
            }
            cc++;
        }
    }
    for (long k = 0; k < len; ++k) {
        const double *vp = V + item[start + k] * r;
        for (int q = 0; q < r; ++q) g[q] = g[q] * 1.0 + vp[q] * t[k];
    }
    /* compute_mm(i, ui, ...) pcr.cpp:549 */
    for (long j = 0; j < len; ++j) {
        const double *vp = V + item[start + j] * r;
        double res = 0.0;
        for (int k = 0; k < r; ++k) res += ui[k] * vp[k];
        mm[j] = res;
    }
    double prev_obj = objective_u_pairs(mm, ui, val + start, len, r, lambda);
    int its = 0, tries = 0;
    if (cc == 0 || norm2(g, r) < 0.0001) {                          /* pcr.cpp:552-559 */
        *obj_u_new = prev_obj;
        memcpy(ui_new, ui, (size_t)r * sizeof(double));
    } else {
        u1ctx_t u = {V, val + start, D, item + start, len, r, lambda};
        its = cg_solve(g, r, hv_u1, &u, delta);
        memcpy(ui_new, ui, (size_t)r * sizeof(double));
        for (int iter = 0; iter < 20; ++iter) {
            for (int q = 0; q < r; ++q) ui_new[q] = ui[q] * 1.0 + delta[q] * -stepsize;
            for (long j = 0; j < len; ++j) {
                const double *vp = V + item[start + j] * r;
                double res = 0.0;
                for (int k = 0; k < r; ++k) res += ui_new[k] * vp[k];
                mm[j] = res;
            }
            *obj_u_new = objective_u_pairs(mm, ui_new, val + start, len, r, lambda);
            ++tries;
            if (*obj_u_new < prev_obj) break;
            else stepsize /= 2.0;
        }
    }
    if (n_ls) *n_ls = tries;
    free(D); free(g); free(delta); free(t); free(mm);
    return its;
}

/* ------------------------------------------------------------------ */
/* drivers pcrpp.cpp:841-901 / pcr.cpp:616-704                         */
/* ------------------------------------------------------------------ */

void orc_train(int solver, long d1, long d2, const long *idx,
               const long *item, const double *val, const long *tidx,
               const long *titem, const double *tval, long tnnz, int r,
               double lambda, int maxiter, int do_predict, int ndcg_k,
               double stepsize, double *U, double *V, orc_iter_t *hist) {
    long nnz = idx[d1];
    double *m = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    double *U_new = (double *)malloc((size_t)(d1 * r) * sizeof(double));
    memset(hist, 0, (size_t)(maxiter + 1) * sizeof(orc_iter_t));
    orc_comp_m(U, V, d1, idx, item, r, m);
    double now_obj = (solver == 2)
        ? orc_objective_new(m, U, V, d1, d2, idx, val, r, lambda)
        : orc_objective(m, U, V, d1, d2, idx, val, r, lambda);
    hist[0].obj = now_obj;
    if (do_predict) {
        orc_eval(U, V, d1, idx, item, val, r, ndcg_k, &hist[0].train_err, &hist[0].train_ndcg);
        if (tnnz != 0) orc_eval(U, V, d1, tidx, titem, tval, r, ndcg_k, &hist[0].test_err, &hist[0].test_ndcg);
    }
    double total_time = 0.0;
    for (int iter = 1; iter <= maxiter; ++iter) {
        double t0 = now_sec();
        int acc = 0, cgv = 0;
        int lsv = update_V_any(solver, d1, d2, idx, item, val, lambda, stepsize, r, U, V, &now_obj, m, &acc, &cgv);
        long tcg = 0, tls = 0;
        if (solver == 2) {
            orc_update_U_new(d1, d2, idx, item, val, m, lambda, stepsize, r, V, U, U_new, &now_obj, &tcg, &tls);
        } else {
            double total = 0.0;                                       /* update_U pcr.cpp:587-611 */
            for (long i = 0; i < d1; ++i) {
                double obj_u = 0.0; int nls = 0;
                tcg += orc_update_u(i, V, idx, item, val, m, r, lambda, stepsize, U + i * r, U_new + i * r, &obj_u, &nls);
                tls += nls;
                total += obj_u;
            }
            total += lambda / 2.0 * norm2_mat(V, d2, r);
            now_obj = total;
        }
        memcpy(U, U_new, (size_t)(d1 * r) * sizeof(double));
        total_time += now_sec() - t0;
        hist[iter].obj = now_obj;
        hist[iter].seconds = total_time;
        hist[iter].cg_v = cgv; hist[iter].ls_v = lsv; hist[iter].cg_u = tcg; hist[iter].ls_u = tls;
        if (do_predict) {
            orc_eval(U, V, d1, idx, item, val, r, ndcg_k, &hist[iter].train_err, &hist[iter].train_ndcg);
            if (tnnz != 0) orc_eval(U, V, d1, tidx, titem, tval, r, ndcg_k, &hist[iter].test_err, &hist[iter].test_ndcg);
        }
    }
    free(m); free(U_new);
}

long orc_count_pairs(long d1, const long *idx, const double *val, int raw) {
    long total = 0;
    for (long i = 0; i < d1; ++i)
        for (long j = idx[i]; j < idx[i + 1]; ++j)
            for (long k = j + 1; k < idx[i + 1]; ++k) {
                if (raw) { if (val[j] != val[k]) total++; }
                else if (lround(val[j]) != lround(val[k])) total++;
            }
    return total;
}
