// pcr_gram.h -- k_ustep_gram: the per-user Newton step of update_u_new (pcrpp.cpp:779-815) for users with FEW ratings
// (n <= 128, below the rank or not far above it), with the small dense object the north star asks for built on the
// matrix cores.
//
// For such a user the dense r x r Hessian H_i = lambda I + V_I^T L V_I is the larger of the two candidate matrices; the
// same Newton step lives in the span of u_i and the user's n rows X = V_I.  Every vector of the CG recurrence
// (pcrpp.cpp:628-647) is kept as   x = alpha * u_i + X^T a   with a scalar alpha and an n-vector a, and with the
// GRAM MATRIX  K = X X^T  (n x n, one MFMA GEMM: v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64) and m = X u_i (the
// scores the sorted state already holds):
//     X x      = alpha m + K a                         (the b of obtain_Hs_new, pcrpp.cpp:592-594 -- no row gather)
//     H x      = lambda x + X^T c,  c = sweep(X x)     -> (lambda alpha, lambda a + c)
//     <x, y>   = ax ay |u|^2 + ax m.a_y + ay m.a_x + a_x . K a_y
// so the whole truncated CG (same alpha / beta / stop formulas, same <= 10 iterations, same windows frozen at the gradient
// point) and every line-search try (m_new = (1 - s alpha_d) m - s K a_d, pcrpp.cpp:728-744) run on n-vectors in LDS; K a is
// kept up to date by the recurrence itself (one matrix-vector product per CG iteration, for the new c).  The rows of V are
// touched TWICE per user -- once by LDS-DMA to form K (the image is then overwritten by K), once at the end for
// u_new = (1 - s alpha_d) u - s X^T a_d -- instead of 1 + 2 n_cg + n_ls ~ 7-10 times.  In exact arithmetic this is the
// reference's step; in floating point it differs from k_ustep by summation order (tests: identical CG / line-search counts
// in fp64, factors to 1e-7).
#pragma once
#include "pcr_kernels.h"

typedef float pcr_f32x16 __attribute__((ext_vector_type(16)));
typedef double pcr_f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct GramMfma;
template <> struct GramMfma<float> {
    static constexpr int TS = 32, KS = 2, NACC = 16;         // tile side, k values per instruction, accumulator registers
    typedef pcr_f32x16 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    // accumulator register j of lane l holds C[row][col]  (cdna_hip_programming.md, fragment layout)
    static __device__ __forceinline__ int crow(int j, int l) { return (j & 3) + 8 * (j >> 2) + 4 * (l >> 5); }
    static __device__ __forceinline__ int ccol(int l) { return l & 31; }
};
template <> struct GramMfma<double> {
    static constexpr int TS = 16, KS = 4, NACC = 4;
    typedef pcr_f64x4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int j, int l) { return (l >> 4) + 4 * j; }
    static __device__ __forceinline__ int ccol(int l) { return l & 15; }
};

// sums of NV values over a team of BLOCK threads; every thread gets all totals.  red: (BLOCK/64) * NV doubles.
template <int BLOCK, int NV>
__device__ __forceinline__ void block_sums(double (&v)[NV], double* red) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (BLOCK == PCR_WAVE) return;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wid * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / PCR_WAVE; ++w) t += red[w * NV + i];
        v[i] = t;
    }
}

// LDS bytes of one workgroup: r-vectors, the X image / K region, the n-vectors
template <typename T>
__host__ __device__ static inline size_t gram_img_bytes(int cap, int nchp) {
    const size_t x = (size_t)cap * nchp * 16, k = (size_t)cap * cap * sizeof(T);
    return x > k ? x : k;
}
template <typename T>
static inline size_t gram_bytes(int cap, int cap_pad, int rs_cap, int ld, int nchp, int block) {
    return 3 * carve_bytes(ld, 8) + carve_bytes((size_t)(block / PCR_WAVE) * 8 + 8, 8) + carve_bytes((size_t)(block / PCR_WAVE) * ld, 8) +
           carve_bytes(gram_img_bytes<T>(cap, nchp), 1) + 12 * carve_bytes(cap, 8) + carve_bytes(cap, sizeof(T)) + carve_bytes(cap_pad, sizeof(T)) +
           carve_bytes(cap, 2) + carve_bytes(cap, 4) + carve_bytes(cap_pad, 4) + carve_bytes(cap + 1, 8) + carve_bytes(rs_cap, 4);
}

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ustep_gram(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                      T* __restrict__ U, const T* __restrict__ Vm, double lambda, double stepsize0,
                                                      int cg_max, double cg_tol, int strict, int solver1, int cap, int cap_pad,
                                                      int rs_cap, int nchp, unsigned long long* counters, int count_rows) {
    typedef GramMfma<T> MM;
    typedef uint32_t LI;
    constexpr int VEC = VecOf<T>::N, NW = BLOCK / PCR_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver cv(smem);
    const int ld = geo.ld;
    double* uvec = cv.take<double>(ld);
    double* unew = cv.take<double>(ld);
    double* tmpv = cv.take<double>(ld);
    double* red = cv.take<double>((size_t)NW * 8 + 8);
    double* wbuf = cv.take<double>((size_t)NW * ld);
    char* img = cv.take<char>(gram_img_bytes<T>(cap, nchp));       // rows of V (16-byte chunks, nchp per row), then K
    double* md = cv.take<double>(cap);     // scores: first the sorted state's (gradient sweep), then m = X u recomputed from the staged rows
    double* ad = cv.take<double>(cap);     // delta
    double* kd = cv.take<double>(cap);
    double* ar = cv.take<double>(cap);     // residual
    double* kr = cv.take<double>(cap);
    double* ap = cv.take<double>(cap);     // direction
    double* kp = cv.take<double>(cap);
    double* ah = cv.take<double>(cap);     // H p
    double* kh = cv.take<double>(cap);
    double* cc = cv.take<double>(cap);     // sweep coefficients
    double* kc = cv.take<double>(cap);
    double* bb = cv.take<double>(cap);     // X p
    T* ms0 = cv.take<T>(cap);
    T* key = cv.take<T>(cap_pad);
    uint16_t* lv0 = cv.take<uint16_t>(cap);
    int32_t* itm = cv.take<int32_t>(cap);
    LI* li = cv.take<LI>(cap_pad);
    double* Sx = cv.take<double>(cap + 1);
    int* rs = cv.take<int>(rs_cap);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lstride = nchp * VEC;
    T* X = reinterpret_cast<T*>(img);
    T* K = reinterpret_cast<T*>(img);

#ifdef PCR_USTEP_PROF
    long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
    const long long tstart_ = tprev_;
#endif
    for (int ui = blockIdx.x; ui < nusers; ui += gridDim.x) {
        const int u = users[ui];
        const int64_t s0 = S.uptr[u];
        const int n = (int)(S.uptr[u + 1] - s0);
        const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
        for (int t = tid; t < ld; t += BLOCK) { uvec[t] = (double)U[(size_t)u * ld + t]; unew[t] = uvec[t]; }
        for (int p = tid; p < n; p += BLOCK) { ms0[p] = S.ms[s0 + p]; md[p] = (double)ms0[p]; lv0[p] = S.slvl[s0 + p]; itm[p] = S.sitem[s0 + p]; }
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = S.runstart[S.runofs[u] + l];
        __syncthreads();
        UPROF(0);
        if (n > 0) stage_rows<T, BLOCK>(Vm, itm, 0, n, X, geo, nchp);             // lands while the gradient sweep runs
        // ---- gradient coefficients (obtain_g_u_new, pcrpp.cpp:506-535): g = lambda u + X^T c
        block_excl_scan<BLOCK>([&](int i) { return md[i]; }, Sx, n, red);
        const bool win = S.ws != 0;
        for (int p = tid; p < n; p += BLOCK)
            cc[p] = win ? sweep_coeff_cached<T>(S, (size_t)s0 + p, Sx, rs, nlev, lv0[p], md[p], 1.0)
                        : sweep_coeff<T>(ms0, Sx, rs, nlev, lv0[p], ms0[p], md[p], 1.0, strict);
        UPROF(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the LDS-DMA of stage_rows
        __syncthreads();
        UPROF(7);
        // m = X u from the rows just staged -- NOT the scores of the sorted state, which are those of the last TRIED V_new when the
        // V step's line search failed (quirk q5: the reference computes the gradient coefficients from that stale m, as the sweep
        // above did, but every b = V_I s and every line-search score from the V it kept, pcrpp.cpp:592-594, :728-744)
        for (int p = tid; p < n; p += BLOCK) {
            const T* xr = X + (size_t)p * lstride;
            double acc0 = 0.0;
            for (int t = 0; t < ld; ++t) acc0 += (double)xr[t] * uvec[t];
            md[p] = acc0;
        }
        // ---- K = X X^T on the matrix cores.  Tile pairs (I <= J) are dealt to the waves; a lane's k range is a contiguous
        // piece of its row (any consistent permutation of k serves a dot product), so it streams through LDS in order.
        const int nt = (n + MM::TS - 1) / MM::TS, npairs = nt * (nt + 1) / 2;
        constexpr int MAXP = (sizeof(T) == 4) ? 3 : 10;        // pairs per wave: fp32 n <= 128 on 4 waves (or 64 on one), fp64 n <= 64
        typename MM::acc_t acc[MAXP];
        int pI[MAXP], pJ[MAXP];
        {
            const int Lk = ld / MM::KS, row = lane % MM::TS, kh = lane / MM::TS;
#pragma unroll
            for (int q = 0; q < MAXP; ++q) {
                const int pr = wid + q * NW;
                pI[q] = -1; pJ[q] = -1;
#pragma unroll
                for (int e = 0; e < MM::NACC; ++e) acc[q][e] = (T)0;
                if (pr < npairs) {
                    int I = 0, rem = pr;                       // pr -> (I, J), I <= J, row-major over the upper triangle
                    while (rem >= nt - I) { rem -= nt - I; ++I; }
                    const int J = I + rem;
                    pI[q] = I; pJ[q] = J;
                    const int ra = I * MM::TS + row, rb = J * MM::TS + row;
                    const T* xa = X + (size_t)(ra < n ? ra : 0) * lstride + kh * Lk;
                    const T* xb = X + (size_t)(rb < n ? rb : 0) * lstride + kh * Lk;
                    const T za = ra < n ? (T)1 : (T)0, zb = rb < n ? (T)1 : (T)0;
                    for (int s = 0; s < Lk; ++s) acc[q] = MM::mma(xa[s] * za, xb[s] * zb, acc[q]);
                }
            }
        }
        __syncthreads();                                                           // every wave is done reading X: K may overwrite it
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            if (pI[q] < 0) continue;
            const int col = pJ[q] * MM::TS + MM::ccol(lane);
#pragma unroll
            for (int e = 0; e < MM::NACC; ++e) {
                const int rowg = pI[q] * MM::TS + MM::crow(e, lane);
                if (rowg < n && col < n) { K[(size_t)rowg * n + col] = acc[q][e]; K[(size_t)col * n + rowg] = acc[q][e]; }
            }
        }
        __syncthreads();
        // y = K a (K symmetric: thread i walks column i, conflict-free; a[j] is an LDS broadcast)
        auto matvec = [&](const double* a, double* y) {
            for (int i = tid; i < n; i += BLOCK) {
                double sacc = 0.0;
                for (int j = 0; j < n; ++j) sacc += (double)K[(size_t)j * n + i] * a[j];
                y[i] = sacc;
            }
            __syncthreads();
        };
        UPROF(2);
        matvec(cc, kc);                                                            // K a_g
        UPROF(3);
        double un2 = 0.0;
        for (int t = tid; t < ld; t += BLOCK) un2 += uvec[t] * uvec[t];
        double d3[3] = {un2, 0.0, 0.0};
        for (int p = tid; p < n; p += BLOCK) { d3[1] += md[p] * cc[p]; d3[2] += cc[p] * kc[p]; }
        block_sums<BLOCK, 3>(d3, red);
        un2 = d3[0];
        // |g|^2 (zero for a user without ratings: g = 0, pcrpp.cpp:496-498)
        const double gn2 = (n == 0) ? 0.0 : lambda * lambda * un2 + 2.0 * lambda * d3[1] + d3[2];
        const double prev_obj = lambda / 2.0 * un2 + S.objp[u];
        double obj_new = prev_obj, loss_new = 0.0;
        int n_cg = 0, n_ls = 0;
        const bool skip = (gn2 < 0.0001) || (solver1 && nlev <= 1);                // pcrpp.cpp:787-790; pcr.cpp:552
#ifdef PCR_GRAM_DEBUG
        if (tid == 0 && n == 10) printf("gram u=%d n=%d un2=%g d3=%g %g gn2=%g prev=%g objp=%g md0=%g cc0=%g kc0=%g K00=%g skip=%d\n", u, n, un2, d3[1], d3[2], gn2, prev_obj, S.objp[u], md[0], cc[0], kc[0], (double)K[0], (int)skip);
#endif
        if (!skip) {
            // ---- CG (solve_delta_u_new, pcrpp.cpp:628-647) on (alpha, a) pairs: delta = 0, rr = -g, p = g
            double al_d = 0.0, al_r = -lambda, al_p = lambda;
            for (int p = tid; p < n; p += BLOCK) {
                ad[p] = 0.0; kd[p] = 0.0;
                ar[p] = cc[p] * -1.0; kr[p] = kc[p] * -1.0;
                ap[p] = cc[p]; kp[p] = kc[p];
            }
            const double err = sqrt(gn2) * cg_tol;
            __syncthreads();
            for (int k = 1; k <= cg_max; ++k) {
                for (int p = tid; p < n; p += BLOCK) bb[p] = al_p * md[p] + kp[p];          // b = X p  (pcrpp.cpp:592-594)
                __syncthreads();
                block_excl_scan<BLOCK>([&](int i) { return bb[i]; }, Sx, n, red);
                for (int p = tid; p < n; p += BLOCK)
                    cc[p] = win ? sweep_coeff_cached<T>(S, (size_t)s0 + p, Sx, rs, nlev, lv0[p], bb[p], 0.0)
                                : sweep_coeff<T>(ms0, Sx, rs, nlev, lv0[p], ms0[p], bb[p], 0.0, strict);
                __syncthreads();
                UPROF(4);
                matvec(cc, kc);
                UPROF(3);
                const double al_h = lambda * al_p;                                         // H p = lambda p + X^T c
                double d5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                for (int p = tid; p < n; p += BLOCK) {
                    const double ahp = lambda * ap[p] + cc[p], khp = lambda * kp[p] + kc[p];
                    ah[p] = ahp; kh[p] = khp;
                    d5[0] += md[p] * ahp; d5[1] += md[p] * ap[p]; d5[2] += md[p] * ar[p]; d5[3] += ap[p] * khp; d5[4] += ar[p] * kp[p];
                }
                block_sums<BLOCK, 5>(d5, red);
                ++n_cg;
                const double pHp = al_p * al_h * un2 + al_p * d5[0] + al_h * d5[1] + d5[3];
                const double rp = al_r * al_p * un2 + al_r * d5[1] + al_p * d5[2] + d5[4];
                const double alpha = -1.0 * rp / pHp;
                al_d += alpha * al_p; al_r += alpha * al_h;
                double e3[3] = {0.0, 0.0, 0.0};
                for (int p = tid; p < n; p += BLOCK) {
                    ad[p] += alpha * ap[p]; kd[p] += alpha * kp[p];
                    const double arn = ar[p] + alpha * ah[p], krn = kr[p] + alpha * kh[p];
                    ar[p] = arn; kr[p] = krn;
                    e3[0] += md[p] * arn; e3[1] += arn * krn; e3[2] += arn * kh[p];
                }
                block_sums<BLOCK, 3>(e3, red);
                const double rr2 = al_r * al_r * un2 + 2.0 * al_r * e3[0] + e3[1];
                const double rHp = al_r * al_h * un2 + al_r * d5[0] + al_h * e3[0] + e3[2];
                if (sqrt(rr2 > 0.0 ? rr2 : 0.0) < err) break;
                const double beta = rHp / pHp;
                al_p = al_r * -1.0 + beta * al_p;
                for (int p = tid; p < n; p += BLOCK) { ap[p] = ar[p] * -1.0 + beta * ap[p]; kp[p] = kr[p] * -1.0 + beta * kp[p]; }
                __syncthreads();
                UPROF(5);
            }
            __syncthreads();
            // ---- line search (pcrpp.cpp:794-813): scores of u - s delta from K a_delta, fresh sort, objective
            double g2[2] = {0.0, 0.0};
            for (int p = tid; p < n; p += BLOCK) { g2[0] += md[p] * ad[p]; g2[1] += ad[p] * kd[p]; }
            block_sums<BLOCK, 2>(g2, red);
            double step = stepsize0, step_used = stepsize0;
            const int npad = next_pow2(n);
            for (int it = 0; it < 20; ++it) {
                step_used = step;                                                          // the user gets the LAST TRIED u (:813)
                const double sc = 1.0 - step * al_d;
                const double nn = sc * sc * un2 - 2.0 * step * sc * g2[0] + step * step * g2[1];      // |u - s delta|^2
                for (int p = tid; p < npad; p += BLOCK) {
                    if (p < n) { key[p] = (T)(sc * md[p] - step * kd[p]); li[p] = LiOps<LI>::pack(lv0[p], (unsigned)p); }
                    else { li[p] = LiOps<LI>::pack(0xFFFFu, (unsigned)p); key[p] = (T)0; }
                }
                __syncthreads();
                UPROF(6);
                bitonic_sort<T, LI, BLOCK, false, true>(key, li, npad, n);                    // update_infor_ui (:684-726)
                UPROF(8);
                loss_new = block_objective<T, BLOCK>(key, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, nlev, n, Sx, red, strict);
                obj_new = lambda / 2.0 * nn + loss_new;
                ++n_ls;
                UPROF(9);
                if (obj_new < prev_obj) break;
                step /= 2.0;
            }
            // ---- the second and last pass over the user's rows: u_new = (1 - s alpha_d) u - s X^T a_delta
            {
                const double sc = 1.0 - step_used * al_d;
                for (int t = tid; t < ld; t += BLOCK) unew[t] = sc * uvec[t];
                for (int p = tid; p < n; p += BLOCK) cc[p] = ad[p] * -step_used;
                __syncthreads();
                block_gather_axpy<T, double, BLOCK, false, 4>(Vm, itm, cc, n, unew, wbuf, geo, 0, false);
            }
        }
        __syncthreads();
        // ---- leave the sorted state of (u_new, V) for the next V step, as k_ustep does
        if (!skip) {
            int32_t* stage = reinterpret_cast<int32_t*>(Sx);
            for (int p = tid; p < n; p += BLOCK) stage[p] = S.sidx[s0 + LiOps<LI>::idx(li[p])];
            __syncthreads();
            for (int p = tid; p < n; p += BLOCK) {
                const LI x = li[p];
                const int lev = (int)LiOps<LI>::lev(x);
                S.ms[s0 + p] = key[p];
                S.slvl[s0 + p] = (uint16_t)lev;
                S.sitem[s0 + p] = itm[LiOps<LI>::idx(x)];
                S.sidx[s0 + p] = stage[p];
                if (S.ws) store_windows<T>(S, (size_t)s0 + p, key, rs, nlev, lev, key[p], strict);
            }
        }
        for (int t = tid; t < ld; t += BLOCK) U[(size_t)u * ld + t] = (T)unew[t];
        if (tid == 0) {
            S.objr[u] = obj_new;
            if (!skip) S.objp[u] = loss_new;
            if (n_cg) atomicAdd(counters + 0, (unsigned long long)n_cg);
            if (n_ls) atomicAdd(counters + 1, (unsigned long long)n_ls);
            if (count_rows) atomicAdd(counters + 2, (unsigned long long)n * (unsigned long long)(skip ? 0 : 2));
        }
        __syncthreads();
        UPROF(10);
        (void)tmpv;
    }
#ifdef PCR_USTEP_PROF
    if (threadIdx.x == 0) {
        const int cls = BLOCK == 64 ? 0 : 1;
        for (int ph = 0; ph < 11; ++ph) atomicAdd(counters + 4 + cls * 16 + ph, (unsigned long long)prof_[ph]);
        atomicAdd(counters + 4 + cls * 16 + 11, (unsigned long long)(clock64() - tstart_));
        atomicAdd(counters + 4 + cls * 16 + 12, 1ull);
    }
#endif
}
