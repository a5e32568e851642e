// pcr_newton.h -- k_unewton: the EXACT per-user Newton direction from the explicit r x r Hessian ("exact-Newton U-step
// with explicit r x r Hessian on MFMA + batched Cholesky"; replaces the truncated CG of solve_delta_u_new, pcrpp.cpp:628-647, and its
// matrix-free obtain_Hs_new, :576-625, for the users it covers).  Optional (pcr_tune "ustep_newton"): the default U step is the
// reference's truncated CG (k_ustep), whose trajectory this mode leaves on purpose.
//
// For one user, at the gradient point (the sorted state k_prepare / k_ustep left: scores, levels, items in (level, m) order):
//     H = lambda I + 2 sum over active pairs (j, q) of (x_j - x_q)(x_j - x_q)^T  =  lambda I + 2 X^T Y,
//     Y_p = deg_p x_p - w_p,   w_p = sum of the rows of p's active partners,   deg_p = their number,
// and the active partners of p inside every other level are one CONTIGUOUS range of that level's run (the windows of the sweeps:
// scores within 1 above / below, pcrpp.cpp:218,224), so w_p is a handful of differences of a prefix table P of the user's rows in
// sorted order.  One workgroup per user (256 threads):
//     1. gradient coefficients c_p and degrees by the same window searches as the sweeps (sweep_coeff, pcr_prims.h);
//     2. one pass over the user's rows of V: the prefix table P (global scratch, L2-resident) and g = lambda u + X^T c;
//     3. a second pass in chunks of 8 rows: X chunk and Y chunk staged in LDS (fp64), H += X^T Y on the matrix cores
//        (v_mfma_f64_16x16x4_f64; the upper triangle of 16 x 16 tiles dealt to the four waves);
//     4. Cholesky of H in LDS, two triangular solves: delta = H^-1 g  -> dir[u];
// k_ustep then takes delta instead of running its CG (line search, re-sorts, objective and the state hand-over are its own).
// Covered: users of at most NEWTON_MAX_N ratings at ranks of at most 112 (H, the per-rating arrays and the staging chunks share the
// 160 KB of LDS); every other user keeps k_ustep's CG, run to convergence (r iterations, tolerance 1e-12) -- the same Newton step by
// another route.  A Hessian that is not positive definite to rounding leaves NaN in dir[u][0], at which k_ustep falls back likewise.
#pragma once
#include "pcr_kernels.h"
#include "pcr_gram.h"

#define NEWTON_MAX_N 1024
#define NEWTON_MAX_LD 112
#define NEWTON_CHUNK 8

// LDS bytes of one workgroup
template <typename T>
static inline size_t newton_bytes(int cap, int rs_cap, int ld, int ldp) {
    return carve_bytes(cap, sizeof(T)) + carve_bytes(cap + 1, 8) + carve_bytes(cap, 2) + carve_bytes(cap, 4) + carve_bytes(cap, 4) +
           carve_bytes(cap, 8) + carve_bytes(rs_cap, 4) + carve_bytes(8, 8) + 3 * carve_bytes(ldp, 8) + carve_bytes(2 * ldp, 8) +
           2 * carve_bytes((size_t)NEWTON_CHUNK * ldp, 8) + carve_bytes((size_t)ld * ld, 8);
}
// + the window bounds of every rating (8 slots of 16 bits), kept where the 160 KB allow it: the Hessian pass then issues all its
// prefix-table loads at once instead of one binary search at a time
static inline size_t newton_wb_bytes(int cap) { return carve_bytes((size_t)cap * 8, 2); }

template <typename T>
__global__ __launch_bounds__(256) void k_unewton(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                 const T* __restrict__ U, const T* __restrict__ Vm, double lambda, int strict,
                                                 int cap, int rs_cap, int ldp, char* scratch, size_t stride, double* __restrict__ dir, int wbcap) {
    constexpr int BLOCK = 256;
    typedef GramMfma<double> MM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver cv(smem);
    const int ld = geo.ld;
    T* ms0 = cv.take<T>(cap);
    double* Sx = cv.take<double>(cap + 1);
    uint16_t* lv0 = cv.take<uint16_t>(cap);
    int32_t* itm = cv.take<int32_t>(cap);
    int32_t* deg = cv.take<int32_t>(cap);
    double* cg = cv.take<double>(cap);
    int* rs = cv.take<int>(rs_cap);
    double* red = cv.take<double>(8);
    double* gv = cv.take<double>(ldp);
    double* yv = cv.take<double>(ldp);
    double* uv = cv.take<double>(ldp);
    double* tot = cv.take<double>(2 * ldp);
    double* Xc = cv.take<double>((size_t)NEWTON_CHUNK * ldp);
    double* Yc = cv.take<double>((size_t)NEWTON_CHUNK * ldp);
    double* Hm = cv.take<double>((size_t)ld * ld);
    uint16_t* wb = cv.take<uint16_t>((size_t)wbcap * 8);          // (wbcap = cap or 0)
    double* Ptab = reinterpret_cast<double*>(scratch + (size_t)blockIdx.x * stride);          // (n + 1) x ldp
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nt = ldp / MM::TS, npairs = nt * (nt + 1) / 2;
    constexpr int MAXP = 7;                                      // 28 tile pairs (nt = 7) over 4 waves

#ifdef PCR_NEWTON_PROF
    long long prof_[6] = {0, 0, 0, 0, 0, 0}, tprev_ = clock64();
#define NPROF(ph) do { if (threadIdx.x == 0) { const long long now_ = clock64(); prof_[ph] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define NPROF(ph) do { } while (0)
#endif
    for (int ui = blockIdx.x; ui < nusers; ui += gridDim.x) {
        const int u = users[ui];
        const int64_t s0 = S.uptr[u];
        const int n = (int)(S.uptr[u + 1] - s0);
        const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
        if (n == 0 || n > cap) continue;                         // (uniform over the workgroup)
        for (int p = tid; p < n; p += BLOCK) { ms0[p] = S.ms[s0 + p]; lv0[p] = S.slvl[s0 + p]; itm[p] = S.sitem[s0 + p]; }
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = S.runstart[S.runofs[u] + l];
        for (int t = tid; t < ldp; t += BLOCK) uv[t] = t < ld ? (double)U[(size_t)u * ld + t] : 0.0;
        __syncthreads();
        // ---- 1. gradient coefficients (obtain_g_u_new, pcrpp.cpp:506-535) and the number of active partners of every rating
        block_excl_scan<BLOCK>([&](int i) { return (double)ms0[i]; }, Sx, n, red);
        for (int p = tid; p < n; p += BLOCK) {
            const T mp = ms0[p];
            const int lev = lv0[p];
            cg[p] = sweep_coeff<T>(ms0, Sx, rs, nlev, lev, mp, (double)mp, 1.0, strict);
            const T lo = mp - (T)1, hi = mp + (T)1;
            int d = 0;
            const bool keep = wbcap > 0 && nlev <= 9;
            for (int l = 0; l < nlev; ++l) {
                if (l == lev) continue;
                const int s = rs[l], e = rs[l + 1];
                int w;
                if (l < lev) { w = strict ? ubound(ms0, s, e, lo) : lbound(ms0, s, e, lo); d += e - w; }
                else { w = strict ? lbound(ms0, s, e, hi) : ubound(ms0, s, e, hi); d += w - s; }
                if (keep) wb[(size_t)p * 8 + (l - (l > lev))] = (uint16_t)w;
            }
            deg[p] = d;
        }
        __syncthreads();
        NPROF(0);
        // ---- 2. prefix table of the rows in sorted order (P[p] = sum of the rows before p) and g = lambda u + X^T c: two halves of
        // the positions side by side (threads [0, 128) / [128, 256), one column each), the second half lifted afterwards
        {
            const int seg = tid >> 7, c = tid & 127, h = n / 2;
            const int p0 = seg ? h : 0, p1 = seg ? n : h;
            double run = 0.0, gp = 0.0;
            if (c < ldp) {
                for (int p = p0; p < p1; p += 4) {
                    double x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = (p + e < p1 && c < ld) ? (double)Vm[(size_t)itm[p + e] * ld + c] : 0.0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (p + e < p1) { Ptab[(size_t)(p + e) * ldp + c] = run; run += x[e]; gp += cg[p + e] * x[e]; }
                }
                tot[seg * ldp + c] = run;
                if (seg) Ptab[(size_t)n * ldp + c] = run;         // row n of the second half (lifted below)
            }
            __syncthreads();
            if (c < ldp) {
                if (seg) {
                    const double base = tot[c];
                    for (int p = p0; p <= n; ++p) Ptab[(size_t)p * ldp + c] += base;
                    yv[c] = gp;                                   // the second half's share of X^T c
                }
            }
            __syncthreads();
            if (!seg && c < ldp) gv[c] = lambda * uv[c] + gp + yv[c];
        }
        __syncthreads();
        NPROF(1);
        // ---- 3. H = lambda I + 2 X^T Y on the matrix cores, 8 rows at a time
        typename MM::acc_t acc[MAXP];
#pragma unroll
        for (int q = 0; q < MAXP; ++q)
#pragma unroll
            for (int e = 0; e < MM::NACC; ++e) acc[q][e] = 0.0;
        const int row = lane % MM::TS, kh = lane / MM::TS;       // MFMA operand indices of this lane (cdna_hip_programming.md)
        for (int c0 = 0; c0 < n; c0 += NEWTON_CHUNK) {
            {   // stage: 32 threads per row of the chunk, up to four columns each; the windows of the row are searched once
                const int pp = tid >> 5, t32 = tid & 31, p = c0 + pp;
                const bool live = p < n;
                double x[4] = {0.0, 0.0, 0.0, 0.0}, w[4] = {0.0, 0.0, 0.0, 0.0};
                if (live) {
                    const T mp = ms0[p];
                    const int lev = lv0[p];
                    const T lo = mp - (T)1, hi = mp + (T)1;
                    const T* vrow = Vm + (size_t)itm[p] * ld;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const int c = t32 + 32 * k; x[k] = c < ld ? (double)vrow[c] : 0.0; }
                    const bool kept = wbcap > 0 && nlev <= 9;
                    for (int l = 0; l < nlev; ++l) {
                        if (l == lev) continue;
                        const int s = rs[l], e = rs[l + 1];
                        int a, b;
                        if (kept) { const int w = wb[(size_t)p * 8 + (l - (l > lev))]; if (l < lev) { a = w; b = e; } else { a = s; b = w; } }
                        else if (l < lev) { a = strict ? ubound(ms0, s, e, lo) : lbound(ms0, s, e, lo); b = e; }
                        else { a = s; b = strict ? lbound(ms0, s, e, hi) : ubound(ms0, s, e, hi); }
                        if (b > a) {
                            const double* pb = Ptab + (size_t)b * ldp;
                            const double* pa = Ptab + (size_t)a * ldp;
#pragma unroll
                            for (int k = 0; k < 4; ++k) { const int c = t32 + 32 * k; if (c < ldp) w[k] += pb[c] - pa[c]; }
                        }
                    }
                }
                const double dg = live ? (double)deg[p] : 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = t32 + 32 * k;
                    if (c < ldp) { Xc[pp * ldp + c] = x[k]; Yc[pp * ldp + c] = dg * x[k] - w[k]; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXP; ++q) {
                const int pr = wid + q * 4;
                if (pr < npairs) {
                    int I = 0, rem = pr;                           // pr -> (I, J), I <= J, row-major over the upper triangle
                    while (rem >= nt - I) { rem -= nt - I; ++I; }
                    const int J = I + rem;
#pragma unroll
                    for (int s = 0; s < NEWTON_CHUNK / MM::KS; ++s) {
                        const int p = s * MM::KS + kh;
                        acc[q] = MM::mma(Xc[p * ldp + I * MM::TS + row], Yc[p * ldp + J * MM::TS + row], acc[q]);
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int pr = wid + q * 4;
            if (pr < npairs) {
                int I = 0, rem = pr;
                while (rem >= nt - I) { rem -= nt - I; ++I; }
                const int J = I + rem;
                const int col = J * MM::TS + MM::ccol(lane);
#pragma unroll
                for (int e = 0; e < MM::NACC; ++e) {
                    const int rw = I * MM::TS + MM::crow(e, lane);
                    if (rw < ld && col < ld) {
                        const double v = 2.0 * acc[q][e] + (rw == col ? lambda : 0.0);
                        // (a diagonal tile holds both triangles; off the diagonal the mirror image is this tile's transpose)
                        if (I != J || rw <= col) { Hm[(size_t)rw * ld + col] = v; Hm[(size_t)col * ld + rw] = v; }
                    }
                }
            }
        }
        __syncthreads();
        NPROF(2);
        // ---- 4. Cholesky (lower triangle, in place): r dependent steps; in the trailing update a lane owns a COLUMN (neighbouring
        // lanes, neighbouring words: no bank conflicts, L[i][k] is a broadcast), the four waves take every fourth row of it, four
        // rows in flight per thread (the loop is LDS-latency-bound: 59 % of the kernel before the rows were spread and unrolled)
        bool bad = false;
        for (int k = 0; k < ld; ++k) {
            const double dkk = Hm[(size_t)k * ld + k];
            if (!(dkk > 0.0)) { bad = true; break; }              // (uniform: every thread reads the same word, behind a barrier)
            const double dk = sqrt(dkk);
            __syncthreads();
            for (int i = k + tid; i < ld; i += BLOCK) Hm[(size_t)i * ld + k] = (i == k) ? dk : Hm[(size_t)i * ld + k] / dk;
            __syncthreads();
            for (int j = k + 1 + lane; j < ld; j += 64) {
                const double ljk = Hm[(size_t)j * ld + k];
                int i = j + wid;
                for (; i + 12 < ld; i += 16) {
                    double h[4], l4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = Hm[(size_t)(i + 4 * e) * ld + j]; l4[e] = Hm[(size_t)(i + 4 * e) * ld + k]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) Hm[(size_t)(i + 4 * e) * ld + j] = h[e] - l4[e] * ljk;
                }
                for (; i < ld; i += 4) Hm[(size_t)i * ld + j] -= Hm[(size_t)i * ld + k] * ljk;
            }
            __syncthreads();
        }
        // ---- the two triangular solves, H delta = g: one wave, ordered by wave-level points only (the LDS serves a wave's accesses
        // in program order)
        if (wid == 0 && !bad) {
            for (int t = lane; t < ld; t += 64) yv[t] = gv[t];
            wave_sync();
            for (int i = 0; i < ld; ++i) {                         // L y = g
                const double yi = yv[i] / Hm[(size_t)i * ld + i];
                wave_sync();
                if (lane == 0) yv[i] = yi;
                for (int j = i + 1 + lane; j < ld; j += 64) yv[j] -= Hm[(size_t)j * ld + i] * yi;
                wave_sync();
            }
            for (int i = ld - 1; i >= 0; --i) {                    // L^T delta = y (in place)
                const double xi = yv[i] / Hm[(size_t)i * ld + i];
                wave_sync();
                if (lane == 0) yv[i] = xi;
                for (int j = lane; j < i; j += 64) yv[j] -= Hm[(size_t)i * ld + j] * xi;
                wave_sync();
            }
            for (int t = lane; t < ld; t += 64) dir[(size_t)u * ld + t] = yv[t];
        }
        __syncthreads();
        NPROF(3);
    }
#ifdef PCR_NEWTON_PROF
    if (threadIdx.x == 0 && blockIdx.x < 4)
        printf("[k_unewton wg %d] kclk: load+coeffs %lld, prefix %lld, hessian %lld, cholesky+solve %lld\n", (int)blockIdx.x, prof_[0] / 1000, prof_[1] / 1000,
               prof_[2] / 1000, prof_[3] / 1000);
#endif
}
