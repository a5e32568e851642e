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
//     2. one pass over the user's rows of V (eight segments of the positions side by side, 16-byte loads): the prefix table P
//        (global scratch, served by the L2s / the Infinity Cache) and g = lambda u + X^T c; the prefix rows at the run boundaries are
//        summed once per level (Bnd), so that a rating needs ONE table row per other level, not two;
//     3. a second pass in chunks of 16 rows: X chunk and Y chunk staged in LDS (fp64) -- the next chunk's loads in flight meanwhile --
//        H += X^T Y on the matrix cores (v_mfma_f64_16x16x4_f64; the upper triangle of 16 x 16 tiles dealt to the four waves,
//        accumulated in registers, one k-step of all of a wave's tile pairs at a time: independent MFMAs back to back);
//     4. BLOCKED Cholesky of H in LDS (round 6): H as 16 x 16 tiles of its lower triangle, laid OVER the per-rating arrays and staging
//        chunks of steps 1-3 (the workgroup needs max(build, factor) of LDS, not the sum: two workgroups per CU); per tile column k
//        the diagonal tile by one wave, the panel below it by row-parallel substitution in registers, the trailing update on the
//        matrix cores -- 3 barriers per tile column (21 at r = 100) where the column-by-column form had 300; then the two
//        triangular solves by one wave with the vector in registers (v_readlane hand-offs, reciprocal diagonal): delta = H^-1 g -> dir[u];
// k_ustep then takes delta instead of running its CG (line search, re-sorts, objective and the state hand-over are its own).
// Covered: users of at most NEWTON_MAX_N ratings at ranks of at most 112 (78.5 KB of LDS per workgroup at 1024 ratings and r = 100); every other user keeps k_ustep's CG, run to convergence (r iterations, tolerance 1e-12) -- the same Newton step by
// another route.  A Hessian that is not positive definite to rounding leaves NaN in dir[u][0], at which k_ustep falls back likewise.
#pragma once
#include "pcr_kernels.h"
#include "pcr_gram.h"

#define NEWTON_MAX_N 1024
#define NEWTON_MAX_LD 112
#define NEWTON_CHUNK 16
#define NEWTON_TP 17                                  // padded row of a 16 x 16 tile in LDS, doubles (columns are read without bank conflicts)
#define NEWTON_TILE (16 * NEWTON_TP)

// LDS bytes of one workgroup: the r-vectors, then ONE region that holds the per-rating arrays, the window bounds and the staging
// chunks while the Hessian is built (in registers) and the Hessian's lower-triangular tiles while it is factored
template <typename T>
static inline size_t newton_build_bytes(int cap, int rs_cap, int ldp, int wbcap) {
    return carve_bytes(cap, sizeof(T)) + carve_bytes(cap + 1, 8) + carve_bytes(cap, 2) + carve_bytes(cap, 4) + carve_bytes(cap, 4) +
           carve_bytes(std::max<size_t>(cap, (size_t)std::min(rs_cap, 10) * ldp), 8) + carve_bytes(rs_cap, 4) + carve_bytes((size_t)wbcap * 8, 2) +
           2 * carve_bytes((size_t)NEWTON_CHUNK * ldp, 8);
}
static inline size_t newton_tiles_bytes(int ldp) { const int nt = ldp / 16; return carve_bytes((size_t)(nt * (nt + 1) / 2) * NEWTON_TILE, 8); }
template <typename T>
static inline size_t newton_bytes(int cap, int rs_cap, int ld, int ldp, int wbcap) {
    (void)ld;
    return carve_bytes(8, 8) + 3 * carve_bytes(ldp, 8) + std::max(newton_build_bytes<T>(cap, rs_cap, ldp, wbcap), newton_tiles_bytes(ldp));
}

__device__ __forceinline__ double newton_readlane(double v, int l) {      // l: wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void newton_load4(const float* p, double (&x)[4]) {      // 16-byte aligned: ld is a multiple of 4
    const float4 v = *reinterpret_cast<const float4*>(p);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}
__device__ __forceinline__ void newton_load4(const double* p, double (&x)[4]) {
    const double2 a = reinterpret_cast<const double2*>(p)[0], b = reinterpret_cast<const double2*>(p)[1];
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
}
__device__ __forceinline__ int newton_tix(int i, int j) { return i * (i + 1) / 2 + j; }                // lower-triangular tile (i, j), i >= j

template <typename T>
__global__ __launch_bounds__(256, 2) void k_unewton(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                 const T* __restrict__ U, const T* __restrict__ Vm, double lambda, int strict,
                                                 int cap, int rs_cap, int ldp, char* scratch, size_t stride, double* __restrict__ dir, int wbcap) {
    constexpr int BLOCK = 256;
    typedef GramMfma<double> MM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver cv(smem);
    const int ld = geo.ld;
    double* red = cv.take<double>(8);
    double* gv = cv.take<double>(ldp);
    double* uv = cv.take<double>(ldp);
    double* dinv = cv.take<double>(ldp);                          // 1 / L[i][i]
    double* Tl = reinterpret_cast<double*>(cv.p);                 // the Hessian's tiles: OVER everything below (steps 1-3 are done by then)
    T* ms0 = cv.take<T>(cap);
    double* Sx = cv.take<double>(cap + 1);
    uint16_t* lv0 = cv.take<uint16_t>(cap);
    int32_t* itm = cv.take<int32_t>(cap);
    int32_t* deg = cv.take<int32_t>(cap);
    const size_t bnd_rows = rs_cap < 10 ? rs_cap : 10;            // Bnd exists for at most 9 levels (the window bounds' eight slots)
    double* cg = cv.take<double>((size_t)cap > bnd_rows * ldp ? (size_t)cap : bnd_rows * ldp);   // step 2's coefficients, then Bnd (step 3)
    int* rs = cv.take<int>(rs_cap);
    uint16_t* wb = cv.take<uint16_t>((size_t)wbcap * 8);          // (wbcap = cap or 0)
    double* Xc = cv.take<double>((size_t)NEWTON_CHUNK * ldp);
    double* Yc = cv.take<double>((size_t)NEWTON_CHUNK * ldp);
    double* Ptab = reinterpret_cast<double*>(scratch + (size_t)blockIdx.x * stride);          // (n + 1) x ldp
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nt = ldp / MM::TS, npairs = nt * (nt + 1) / 2;
    constexpr int MAXP = 7;                                      // 28 tile pairs (nt = 7) over 4 waves

#ifdef PCR_NEWTON_PROF
    long long prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
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
        // ---- 2. prefix table of the rows in sorted order (P[p] = sum of the rows before p) and g = lambda u + X^T c: EIGHT segments of
        // the positions side by side (32 threads each, four columns per thread: one 16-byte load per row of V), every segment a local
        // prefix first; the segments' totals meet in LDS (over the idle staging chunks) and segments 1..7 are lifted afterwards
        {
            const int seg = tid >> 5, c = 4 * (tid & 31);
            const int p0 = (int)((long long)n * seg / 8), p1 = (int)((long long)n * (seg + 1) / 8);
            double run[4] = {0.0, 0.0, 0.0, 0.0}, gp[4] = {0.0, 0.0, 0.0, 0.0};
            double* segtot = Xc;                                  // 8 x ldp
            double* seggp = Yc;                                   // 8 x ldp
            if (c < ldp) {
                for (int p = p0; p < p1; p += 4) {
                    double x[4][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (p + e < p1 && c < ld) newton_load4(Vm + (size_t)itm[p + e] * ld + c, x[e]);
                        else { x[e][0] = x[e][1] = x[e][2] = x[e][3] = 0.0; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (p + e < p1) {
                            double* pt = Ptab + (size_t)(p + e) * ldp + c;
                            const double ce = cg[p + e];
#pragma unroll
                            for (int q = 0; q < 4; ++q) { pt[q] = run[q]; run[q] += x[e][q]; gp[q] += ce * x[e][q]; }
                        }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { segtot[seg * ldp + c + q] = run[q]; seggp[seg * ldp + c + q] = gp[q]; }
            }
            __syncthreads();
            if (c < ldp) {
                double base[4] = {0.0, 0.0, 0.0, 0.0};
                for (int s2 = 0; s2 < seg; ++s2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) base[q] += segtot[s2 * ldp + c + q];
                if (seg > 0)
                    for (int p = p0; p < p1; ++p) {
                        double* pt = Ptab + (size_t)p * ldp + c;
#pragma unroll
                        for (int q = 0; q < 4; ++q) pt[q] += base[q];
                    }
                if (seg == 7) {
                    double* pt = Ptab + (size_t)n * ldp + c;      // row n: the sum of all rows
#pragma unroll
                    for (int q = 0; q < 4; ++q) pt[q] = base[q] + run[q];
                }
            }
            if (tid < ldp) {
                double g = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) g += seggp[s2 * ldp + tid];
                gv[tid] = lambda * uv[tid] + g;
            }
        }
        __syncthreads();
        // The partners of rating p inside another level l are a window of l's run: rows [w, end of the run) for the levels below p's,
        // [start of the run, w) for those above -- so  w_p = Bnd[level of p] + sum_{l above} P[w_l] - sum_{l below} P[w_l]  with the run
        // boundaries' prefix rows summed ONCE per level (Bnd, over the dead coefficients): one table row per rating and level, not two.
        const bool kept = wbcap > 0 && nlev <= 9;
        double* Bnd = cg;
        if (kept) {
            if (tid < ldp) {
                double run = 0.0;
                for (int l = 0; l < nlev; ++l) { Bnd[l * ldp + tid] = run; run += Ptab[(size_t)rs[l + 1] * ldp + tid]; }
                run = 0.0;
                for (int l = nlev - 1; l >= 0; --l) { Bnd[l * ldp + tid] -= run; run += Ptab[(size_t)rs[l] * ldp + tid]; }
            }
            __syncthreads();
        }
        NPROF(1);
        // ---- 3. H = lambda I + 2 X^T Y on the matrix cores, 8 rows at a time
        typename MM::acc_t acc[MAXP];
#pragma unroll
        for (int q = 0; q < MAXP; ++q)
#pragma unroll
            for (int e = 0; e < MM::NACC; ++e) acc[q][e] = 0.0;
        const int row = lane % MM::TS, kh = lane / MM::TS;       // MFMA operand indices of this lane (cdna_hip_programming.md)
        // this wave's tile pairs (I, J), I <= J, as LDS offsets of its two operands inside a staged row
        int offA[MAXP], offB[MAXP];
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int pr = wid + q * 4;
            int I = 0, rem = pr < npairs ? pr : 0;                 // pr -> (I, J), row-major over the upper triangle
            while (rem >= nt - I) { rem -= nt - I; ++I; }
            offA[q] = I * MM::TS + row; offB[q] = (I + rem) * MM::TS + row;
        }
        // Staging, 16 threads per row of a chunk, up to seven columns each, in two halves so that the NEXT chunk's loads are in flight
        // while this chunk is on the matrix cores: issue() starts the row of V and the first four prefix-table rows (a shard of at
        // most five levels needs no more), consume() turns them into the row's X and Y values after the chunk's MFMAs.
        const int pp = tid >> 4, t16 = tid & 15;
        int st_p = 0, st_lev = 0;
        bool st_live = false;
        double st_dg = 0.0, tt[4][7], sg[4];
        T xr[7];
        auto issue = [&](int c0) {
            st_p = c0 + pp; st_live = st_p < n;
#pragma unroll
            for (int k = 0; k < 7; ++k) xr[k] = (T)0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                sg[q4] = 0.0;
#pragma unroll
                for (int k = 0; k < 7; ++k) tt[q4][k] = 0.0;
            }
            if (st_live) {
                st_lev = lv0[st_p]; st_dg = (double)deg[st_p];
                const T* vrow = Vm + (size_t)itm[st_p] * ld;
#pragma unroll
                for (int k = 0; k < 7; ++k) { const int c = t16 + 16 * k; if (c < ld) xr[k] = vrow[c]; }
                if (kept) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const bool on = q4 < nlev - 1;
                        const double* prow = Ptab + (size_t)(on ? wb[(size_t)st_p * 8 + q4] : 0) * ldp;
                        sg[q4] = !on ? 0.0 : (q4 < st_lev ? -1.0 : 1.0);                 // slot sl = level sl below p's level, level sl + 1 above
#pragma unroll
                        for (int k = 0; k < 7; ++k) if (k < nt) tt[q4][k] = prow[t16 + 16 * k];      // (k < nt: uniform; t16 + 16 k < ldp)
                    }
                }
            }
        };
        auto consume = [&]() {                                     // ... and stores the row into the (idle) staging chunk
            double w[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (st_live) {
                if (kept) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                        for (int k = 0; k < 7; ++k) w[k] += sg[q4] * tt[q4][k];
                    for (int sl = 4; sl < nlev - 1; ++sl) {          // levels six to nine: not prefetched
                        const double* prow = Ptab + (size_t)wb[(size_t)st_p * 8 + sl] * ldp;
                        const double sgn = sl < st_lev ? -1.0 : 1.0;
#pragma unroll
                        for (int k = 0; k < 7; ++k) if (k < nt) w[k] += sgn * prow[t16 + 16 * k];
                    }
#pragma unroll
                    for (int k = 0; k < 7; ++k) if (k < nt) w[k] += Bnd[st_lev * ldp + t16 + 16 * k];
                } else {
                    const T mp = ms0[st_p];
                    const T lo = mp - (T)1, hi = mp + (T)1;
                    for (int l = 0; l < nlev; ++l) {
                        if (l == st_lev) continue;
                        const int s0 = rs[l], e = rs[l + 1];
                        int a2, b2;
                        if (l < st_lev) { a2 = strict ? ubound(ms0, s0, e, lo) : lbound(ms0, s0, e, lo); b2 = e; }
                        else { a2 = s0; b2 = strict ? lbound(ms0, s0, e, hi) : ubound(ms0, s0, e, hi); }
                        if (b2 > a2) {
                            const double* pb = Ptab + (size_t)b2 * ldp;
                            const double* pa = Ptab + (size_t)a2 * ldp;
#pragma unroll
                            for (int k = 0; k < 7; ++k) if (k < nt) w[k] += pb[t16 + 16 * k] - pa[t16 + 16 * k];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < nt) { const int c = t16 + 16 * k; const double xk = (double)xr[k]; Xc[pp * ldp + c] = xk; Yc[pp * ldp + c] = st_dg * xk - w[k]; }
        };
        issue(0);
        consume();
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += NEWTON_CHUNK) {
            const bool more = c0 + NEWTON_CHUNK < n;
            if (more) issue(c0 + NEWTON_CHUNK);
            NPROF(4);
            // one k-step of all seven pairs at a time: fourteen operand reads, then seven independent MFMAs
#pragma unroll
            for (int s4 = 0; s4 < NEWTON_CHUNK / MM::KS; ++s4) {
                const int pb = (s4 * MM::KS + kh) * ldp;
                double av[MAXP], bv[MAXP];
#pragma unroll
                for (int q = 0; q < MAXP; ++q) { av[q] = Xc[pb + offA[q]]; bv[q] = Yc[pb + offB[q]]; }
#pragma unroll
                for (int q = 0; q < MAXP; ++q) acc[q] = MM::mma(av[q], bv[q], acc[q]);
            }
            __syncthreads();
            NPROF(5);
            if (more) { consume(); __syncthreads(); }
        }
        // H = lambda I + 2 X^T Y into the lower-triangular tiles (tile (J, I) is the transpose of the pair (I, J) just accumulated); the
        // tiles lie over the arrays of steps 1-3, which nobody reads any more (the loop above ends in a barrier).  Rows / columns
        // beyond r (the padding to a multiple of 16) get a unit diagonal: the factorisation passes through them unchanged.
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int pr = wid + q * 4;
            if (pr < npairs) {
                int I = 0, rem = pr;
                while (rem >= nt - I) { rem -= nt - I; ++I; }
                const int J = I + rem;
                double* tile = Tl + (size_t)newton_tix(J, I) * NEWTON_TILE;
                const int cl = MM::ccol(lane), col = J * MM::TS + cl;
#pragma unroll
                for (int e = 0; e < MM::NACC; ++e) {
                    const int rl = MM::crow(e, lane), rw = I * MM::TS + rl;
                    double v = 2.0 * acc[q][e] + (rw == col ? lambda : 0.0);
                    if (rw >= ld || col >= ld) v = rw == col ? 1.0 : 0.0;
                    tile[cl * NEWTON_TP + rl] = v;
                }
            }
        }
        __syncthreads();
        NPROF(2);
        // ---- 4. blocked Cholesky of the tiles (lower triangle, in place)
        if (tid == 0) red[7] = 0.0;                                 // "not positive definite" flag
        __syncthreads();
        for (int k = 0; k < nt; ++k) {
            double* D = Tl + (size_t)newton_tix(k, k) * NEWTON_TILE;
            if (wid == 0) {                                        // (a) the diagonal tile: lanes 0..15 hold one ROW each in registers, column by
                double a[16];                                      //     column (left-looking); row kk's finished entries come by v_readlane
                const int rr = lane & 15;
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = D[rr * NEWTON_TP + c];
                bool pd = true;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
                    for (int j = 0; j < kk; ++j) a[kk] -= a[j] * newton_readlane(a[j], kk);
                    const double dkk = newton_readlane(a[kk], kk);
                    pd = pd && dkk > 0.0;                          // (uniform; no way out of the loop: it must unroll, the row lives in registers)
                    const double ri = 1.0 / sqrt(dkk);
                    a[kk] = rr == kk ? dkk * ri : a[kk] * ri;      // (rows above kk hold the unused upper triangle)
                    if (lane == 0) dinv[k * 16 + kk] = ri;
                }
                if (!pd) { if (lane == 0) red[7] = 1.0; }
                else if (lane < 16) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) D[rr * NEWTON_TP + c] = a[c];
                }
            }
            __syncthreads();
            NPROF(6);
            if (red[7] != 0.0) break;                              // (uniform: read behind the barrier)
            const int below = nt - k - 1;
            if (tid < below * 16) {                                // (b) the panel below: X L_kk^T = A, one row per thread, in registers
                double* A = Tl + (size_t)newton_tix(k + 1 + (tid >> 4), k) * NEWTON_TILE + (tid & 15) * NEWTON_TP;
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = A[c];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double sacc = x[c];
#pragma unroll
                    for (int j = 0; j < c; ++j) sacc -= x[j] * D[c * NEWTON_TP + j];
                    x[c] = sacc * dinv[k * 16 + c];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) A[c] = x[c];
            }
            __syncthreads();
            NPROF(7);
            const int tp = below * (below + 1) / 2;                // (c) trailing update A_ij -= L_ik L_jk^T, k < j <= i, on the matrix cores
            for (int pr = wid; pr < tp; pr += 4) {
                int i = 0, rem = pr;                               // pr -> (i, j), j <= i, row-major over the lower triangle
                while (rem > i) { rem -= i + 1; ++i; }
                const int j = rem;
                double* C = Tl + (size_t)newton_tix(k + 1 + i, k + 1 + j) * NEWTON_TILE;
                const double* Li = Tl + (size_t)newton_tix(k + 1 + i, k) * NEWTON_TILE;
                const double* Lj = Tl + (size_t)newton_tix(k + 1 + j, k) * NEWTON_TILE;
                typename MM::acc_t c4;
#pragma unroll
                for (int e = 0; e < MM::NACC; ++e) c4[e] = C[MM::crow(e, lane) * NEWTON_TP + MM::ccol(lane)];
#pragma unroll
                for (int s4 = 0; s4 < 16 / MM::KS; ++s4)
                    c4 = MM::mma(-Li[row * NEWTON_TP + s4 * MM::KS + kh], Lj[row * NEWTON_TP + s4 * MM::KS + kh], c4);
#pragma unroll
                for (int e = 0; e < MM::NACC; ++e) C[MM::crow(e, lane) * NEWTON_TP + MM::ccol(lane)] = c4[e];
            }
            __syncthreads();
            NPROF(8);
        }
        const bool bad = red[7] != 0.0;
        // ---- the two triangular solves, H delta = g: one wave, the vector in registers (lane l holds entries l and l + 64), the
        // pivot entry handed round by v_readlane, L's column / row of the NEXT step loaded while this one's update runs
        if (wid == 0 && !bad) {
            // lane l holds entries l (y0) and l + 64 (y1) of the vector; step i hands y_i / L_ii round by v_readlane and every lane
            // below (forward) / above (backward) takes its share; L's entries of one tile column / row are loaded ahead, from a tile
            // that always exists (a lane whose entry is already final, or beyond ldp, loads a harmless one: its update is masked)
            const int j0 = lane, j1 = lane + 64, t0 = j0 >> 4, t1 = j1 >> 4;
            double y0 = j0 < ldp ? gv[j0] : 0.0, y1 = j1 < ldp ? gv[j1] : 0.0;
            const double dv0 = j0 < ldp ? dinv[j0] : 1.0, dv1 = j1 < ldp ? dinv[j1] : 1.0;
            const int tc0 = t0 < nt ? t0 : nt - 1, tc1 = t1 < nt ? t1 : nt - 1;
            for (int ib = 0; ib < nt; ++ib) {                      // L y = g, one tile column of L at a time
                double l0[16], l1[16];
                const double* q0 = Tl + (size_t)newton_tix(tc0 < ib ? ib : tc0, ib) * NEWTON_TILE + (j0 & 15) * NEWTON_TP;
                const double* q1 = Tl + (size_t)newton_tix(tc1 < ib ? ib : tc1, ib) * NEWTON_TILE + (j1 & 15) * NEWTON_TP;
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) { l0[ii] = q0[ii]; l1[ii] = q1[ii]; }
                if (ib < 4) {
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) {
                        const int i = ib * 16 + ii;
                        const double yi = newton_readlane(y0 * dv0, i);
                        y0 = j0 == i ? yi : (j0 > i ? y0 - l0[ii] * yi : y0);
                        y1 -= l1[ii] * yi;                         // (j1 > i for every i < 64)
                    }
                } else {
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) {
                        const int i = ib * 16 + ii;
                        const double yi = newton_readlane(y1 * dv1, i - 64);
                        y1 = j1 == i ? yi : (j1 > i ? y1 - l1[ii] * yi : y1);
                    }
                }
            }
            for (int ib = nt - 1; ib >= 0; --ib) {                 // L^T delta = y (in place), one tile ROW of L at a time
                double l0[16], l1[16];
                const double* q0 = Tl + (size_t)newton_tix(ib, tc0 > ib ? ib : tc0) * NEWTON_TILE + (j0 & 15);
                const double* q1 = Tl + (size_t)newton_tix(ib, tc1 > ib ? ib : tc1) * NEWTON_TILE + (j1 & 15);
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) { l0[ii] = q0[ii * NEWTON_TP]; l1[ii] = q1[ii * NEWTON_TP]; }
                if (ib >= 4) {
#pragma unroll
                    for (int ii = 15; ii >= 0; --ii) {
                        const int i = ib * 16 + ii;
                        const double xi = newton_readlane(y1 * dv1, i - 64);
                        y1 = j1 == i ? xi : (j1 < i ? y1 - l1[ii] * xi : y1);
                        y0 -= l0[ii] * xi;                         // (j0 < i for every i >= 64)
                    }
                } else {
#pragma unroll
                    for (int ii = 15; ii >= 0; --ii) {
                        const int i = ib * 16 + ii;
                        const double xi = newton_readlane(y0 * dv0, i);
                        y0 = j0 == i ? xi : (j0 < i ? y0 - l0[ii] * xi : y0);
                    }
                }
            }
            if (j0 < ld) dir[(size_t)u * ld + j0] = y0;
            if (j1 < ld) dir[(size_t)u * ld + j1] = y1;
        }
        __syncthreads();
        NPROF(3);
    }
#ifdef PCR_NEWTON_PROF
    if (threadIdx.x == 0 && blockIdx.x < 4)
        printf("[k_unewton wg %d] kclk: load+coeffs %lld, prefix %lld, hessian %lld (staging %lld + mfma %lld + tiles out %lld), cholesky %lld (diagonal %lld + panel %lld + trailing %lld), solves %lld\n", (int)blockIdx.x,
               prof_[0] / 1000, prof_[1] / 1000, (prof_[2] + prof_[4] + prof_[5]) / 1000, prof_[4] / 1000, prof_[5] / 1000, prof_[2] / 1000,
               (prof_[6] + prof_[7] + prof_[8]) / 1000, prof_[6] / 1000, prof_[7] / 1000, prof_[8] / 1000, prof_[3] / 1000);
#endif
}
