// pcr_ustep.h -- k_ustep: the per-user Newton step of update_u_new (pcrpp.cpp:779-815), workgroup clusters for the longest
// users.  Part of pcr_kernels.h.
#pragma once
#include "pcr_prims.h"

// ---------------------------------------------------------------------------------------
// Workgroup clusters for long users.  One workgroup is bound by one CU's gather bandwidth
// (~50-70 GB/s), so a user with thousands of ratings is given K workgroups (on K CUs): every member
// runs the SAME per-user program on the same data (scan, sweep, CG scalars, sort, line-search
// decisions are recomputed redundantly and are bitwise identical, so the members never have to
// agree on control flow), but each member gathers only its 1/K slice of the rows; slices of scores
// and partial r-vectors are exchanged through global memory.
// Hand-off protocol (cdna_hip_programming.md Guideline 16): every handed-off byte is stored by an agent-scope (sc1, written
// through) store -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0: agent-scope RELEASE, vmcnt(0), relaxed agent
// atomic add on the cluster's arrival counter -> relaxed poll (bounded, with s_sleep) -> agent-scope ACQUIRE, vmcnt(0) ->
// workgroup barrier -> loads of the handed-off bytes (agent-scope, sc1).  That is the formally ordered form and the only one.
// (Without the release and the acquire -- the payload is sc1 both ways, the first row of MI355X_MICROARCH.md's table of hand-offs
// measured valid on gfx950, "not an architectural guarantee", and its "one workgroup per CU" cell does not hold while other length
// classes share the CUs -- the cluster class ran 5 % faster, 0.7 % of a step: measured in rounds 2-3, not kept.)
// Placement-independent; the launch keeps the grid <= one workgroup per CU so all members are co-resident.
// ---------------------------------------------------------------------------------------
struct ClusterBufs {
    unsigned* bar;          // one arrival counter per cluster (zeroed before every launch)
    char* xch;              // per cluster: 2 x cap_pad scores (T) + 2 x K x ld doubles
    size_t xch_stride;
    unsigned long long* rows;   // this length class's cumulative count of gathered rows (pcr_tune "count_rows"; never reset)
};

template <int K>
__device__ __forceinline__ void cluster_barrier(unsigned* bar, unsigned& phase, unsigned long long* err) {
    if (K == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its write-through stores
    __syncthreads();
    phase += 1;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (explicit: the compiler may drop the wait behind the write-back)
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = phase * K;
        unsigned spins = 0;
        // bounded wait: a cluster that lost a member reports an error instead of hanging the GPU,
        // and once any cluster has failed nobody waits any more
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
            if (spins > (1u << 21)) { atomicAdd(err, 1ull); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");            // (no instruction: keeps the payload loads below the poll)
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// k_ustep: the whole per-user Newton step of update_u_new (pcrpp.cpp:779-815) in one
// workgroup: gradient (obtain_g_u_new :493), objective (:542), <=10 CG iterations with
// obtain_Hs_new (:576, :628), <=20 line-search evaluations each with a fresh sort (:794-813).
// r-vectors live in LDS as fp64; the user's sorted item block lives in LDS (or scratch).
// ---------------------------------------------------------------------------------------
template <typename T>
static inline size_t ustep_big_bytes(int cap, int cap_pad, int rs_cap, int li_bytes) {
    return carve_bytes(cap, sizeof(T)) + carve_bytes(cap_pad, sizeof(T)) + carve_bytes(cap, 2) + carve_bytes(cap, 4) +
           carve_bytes(cap_pad, li_bytes) + carve_bytes(cap + 1, 8) + carve_bytes(rs_cap, 4);
}
static inline size_t ustep_small_bytes(int ld, int block, size_t elt) {
    return carve_bytes(ld, elt) + carve_bytes(block / PCR_WAVE + 1, 8) + 8 * carve_bytes(ld, 8) +
           carve_bytes((size_t)(block / PCR_WAVE) * ld, 8);
}
static inline size_t ustep_rows_bytes(int rcap, int nchp) { return (size_t)rcap * nchp * 16; }
template <typename T>
static inline size_t ustep_xch_bytes(int cap_pad, int ld, int K) {
    return 2 * carve_bytes(cap_pad, sizeof(T)) + 2 * carve_bytes((size_t)K * ld, 8);
}

#ifdef PCR_USTEP_PROF
#define UPROF(ph) do { if (threadIdx.x == 0) { const long long now_ = clock64(); prof_[ph] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define UPROF(ph) do { } while (0)
#endif
// RES: the workgroup keeps rows of V in LDS (rcap > 0); UNR: rows in flight per lane group of the L2 gathers (8 for the
// latency-bound classes with few users, one workgroup per CU; 4 keeps the kernel at <= 128 VGPRs so that two 512-thread
// workgroups share a CU in the throughput-bound classes with many users).
// CLS: nothing but a distinct kernel SYMBOL for two length classes that run the same workgroup form, so that a profiler's
// per-symbol figures (rocprofv3 --stats, --pmc) belong to one class each.
// The 512-thread throughput form (4 rows in flight, no LDS image, no cluster) must stay within 128 VGPRs = 4 waves per SIMD, so
// that two workgroups share a CU: its CLS = 0 symbol is compiled under HIP's minimum-waves-per-SIMD bound (the second
// __launch_bounds__ argument), which also caps the dynamic LDS a launch may ask for at half a CU's -- a class whose
// per-rating arrays need more than that runs one workgroup per CU whatever its registers and takes the CLS = 1 symbol.
template <typename T, int BLOCK, bool BIG, int K, bool RES, int UNR, int CLS = 0>
__global__ __launch_bounds__(BLOCK, (BLOCK == 512 && UNR == 4 && !RES && K == 1 && !BIG && CLS == 0 && sizeof(T) == 4) ? 4 : 1) void k_ustep(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                 T* __restrict__ U, const T* __restrict__ Vm, double lambda, double stepsize0,
                                                 int cg_max, double cg_tol, int strict, int solver1, int cap, int cap_pad, int rs_cap, int rcap, int nchp,
                                                 char* scratch, size_t stride, unsigned long long* counters, ClusterBufs cb, int fault,
                                                 int wcap, const double* __restrict__ dir = nullptr) {
    typedef typename LiSel<T, BIG>::type LI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // test hook (pcr_tune "fault_cluster_member"): the last member of every cluster leaves at once, so that the others run
    // into the bounded wait of cluster_barrier and the launch reports a time-out instead of hanging
    if (K > 1 && (fault & 1) && (int)(blockIdx.x % K) == K - 1) return;
    Carver small(smem);
    T* vecT = small.take<T>(geo.ld);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    double* uvec = small.take<double>(geo.ld);
    double* gvec = small.take<double>(geo.ld);
    double* delta = small.take<double>(geo.ld);
    double* rr = small.take<double>(geo.ld);
    double* pv = small.take<double>(geo.ld);
    double* Hp = small.take<double>(geo.ld);
    double* unew = small.take<double>(geo.ld);
    double* part = small.take<double>(geo.ld);
    double* wbuf = small.take<double>((size_t)(BLOCK / PCR_WAVE) * geo.ld);
    // LDS image of the first rcap rows of V this workgroup gathers for its user (stage_rows): every pass of the Newton
    // step over those rows (gradient, 2 per CG iteration, 1 per line-search try) reads LDS instead of L2
    constexpr int VEC = VecOf<T>::N;
    const int lstride = nchp * VEC;
    T* rowsL = small.take<T>((size_t)rcap * lstride);
    // LDS copy of the user's window rows (16-bit: LDS-resident users have fewer than 65536 ratings): the gradient sweep and
    // every CG sweep read them -- from global memory that is one dependent round trip per sweep, ~2.5 us each while the other
    // length classes keep the memory pipe busy (wcap = 0: no copy, e.g. the global-scratch classes)
    uint16_t* winL = small.take<uint16_t>((size_t)wcap);
    Carver big(BIG ? scratch + (size_t)blockIdx.x * stride : small.p);
    T* ms0 = big.take<T>(cap);
    T* key = big.take<T>(cap_pad);
    uint16_t* lv0 = big.take<uint16_t>(cap);
    int32_t* itm = big.take<int32_t>(cap);
    LI* li = big.take<LI>(cap_pad);
    double* Sx = big.take<double>(cap + 1);
    int* rs = big.take<int>(rs_cap);
    const int tid = threadIdx.x;
    const int ld = geo.ld;
    // cluster geometry: member j of cluster cid gathers rows [r0, r1) of every user it works on
    const int cid = blockIdx.x / K, mem = blockIdx.x % K, nclus = gridDim.x / K;
    unsigned phase = 0, xs_par = 0, xv_par = 0;
    unsigned* bar = (K > 1) ? cb.bar + cid : nullptr;
    T *xs0 = nullptr, *xs1 = nullptr;
    double *xv0 = nullptr, *xv1 = nullptr;
    if (K > 1) {
        Carver xc(cb.xch + (size_t)cid * cb.xch_stride);
        xs0 = xc.take<T>(cap_pad); xs1 = xc.take<T>(cap_pad);
        xv0 = xc.take<double>((size_t)K * ld); xv1 = xc.take<double>((size_t)K * ld);
    }
    // all members end up with the full score vector in key[0, n)
    auto exchange_scores = [&](T* key, int n, int r0, int r1) {
        if (K == 1) return;
        T* buf = (xs_par & 1) ? xs1 : xs0; xs_par += 1;
        // exchange buffers are re-used, and the per-XCD L2s are not coherent with each other: every
        // store and load of handed-off bytes is agent-scope (sc1: write-through / L2-revalidated)
        for (int p = r0 + tid; p < r1; p += BLOCK) __hip_atomic_store(buf + p, key[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cluster_barrier<K>(bar, phase, counters + 3);
        for (int p = tid; p < n; p += BLOCK) key[p] = __hip_atomic_load(buf + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    };
    // vec += sum over members (fixed order) of their partial r-vectors
    auto exchange_vector = [&](double* vec) {
        if (K == 1) return;
        double* buf = (xv_par & 1) ? xv1 : xv0; xv_par += 1;
        for (int t = tid; t < ld; t += BLOCK)
            __hip_atomic_store(buf + (size_t)mem * ld + t, part[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cluster_barrier<K>(bar, phase, counters + 3);
        for (int t = tid; t < ld; t += BLOCK) {
            double sum = 0.0;
            for (int j = 0; j < K; ++j) sum += __hip_atomic_load(buf + (size_t)j * ld + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vec[t] += sum;
        }
        __syncthreads();
    };

#ifdef PCR_USTEP_PROF
    long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
    const long long tstart_ = tprev_;
#endif
    for (int ui = cid; ui < nusers; ui += nclus) {
        const int u = users[ui];
        const int64_t s0 = S.uptr[u];
        const int n = (int)(S.uptr[u + 1] - s0);
        const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
        const int r0 = (int)((int64_t)n * mem / K), r1 = (int)((int64_t)n * (mem + 1) / K);
        const int q0 = r0, q1 = RES ? min(r1, r0 + rcap) : r0;       // rows [q0, q1) are LDS-resident, [q1, r1) stay in L2
        // out[p] = vec . V[item p] over this member's rows
        auto sddmm = [&](T* out) {
            if (RES && q1 > q0) block_sddmm<T, BLOCK, true, UNR>(rowsL, vecT, nullptr, q1, out, geo, q0, lstride);
            if (r1 > q1) block_sddmm<T, BLOCK, false, UNR>(Vm, vecT, itm, r1, out, geo, q1);
        };
        // vec += sum_p c[p] V[item p] over all rows of the user (cluster: partials exchanged)
        auto gather_axpy = [&](const T* c, double* vec) {
            if (K == 1) {
                if (RES && q1 > q0) block_gather_axpy<T, T, BLOCK, true, UNR>(rowsL, nullptr, c, q1, vec, wbuf, geo, q0, false, lstride);
                if (r1 > q1) block_gather_axpy<T, T, BLOCK, false, UNR>(Vm, itm, c, r1, vec, wbuf, geo, q1, false);
                if (r1 == q0) __syncthreads();
            } else {
                if (RES && q1 > q0) block_gather_axpy<T, T, BLOCK, true, UNR>(rowsL, nullptr, c, q1, part, wbuf, geo, q0, true, lstride);
                if (r1 > q1 || q1 == q0) block_gather_axpy<T, T, BLOCK, false, UNR>(Vm, itm, c, r1, part, wbuf, geo, q1, q1 == q0);
                exchange_vector(vec);
            }
        };
        for (int t = tid; t < ld; t += BLOCK) uvec[t] = (double)U[(size_t)u * ld + t];
#pragma unroll 4
        for (int p = tid; p < n; p += BLOCK) { ms0[p] = S.ms[s0 + p]; lv0[p] = S.slvl[s0 + p]; itm[p] = S.sitem[s0 + p]; }
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = S.runstart[S.runofs[u] + l];
        const bool win = S.ws != 0;                                             // windows of the gradient point are cached
        const bool wl = win && wcap >= n * S.ws;
        if (wl) {
            if (S.w16) { const uint16_t* wg = reinterpret_cast<const uint16_t*>(S.win) + (size_t)s0 * S.ws; for (int i = tid; i < n * S.ws; i += BLOCK) winL[i] = wg[i]; }
            else { const uint32_t* wg = reinterpret_cast<const uint32_t*>(S.win) + (size_t)s0 * S.ws; for (int i = tid; i < n * S.ws; i += BLOCK) winL[i] = (uint16_t)wg[i]; }
        }
        __syncthreads();
        if (RES && q1 > q0) stage_rows<T, BLOCK>(Vm, itm, q0, q1, rowsL, geo, nchp);      // lands while the gradient sweep runs
        UPROF(0);
        // ---- gradient coefficients, obtain_g_u_new (pcrpp.cpp:506-535)
        block_excl_scan<BLOCK>([&](int i) { return (double)ms0[i]; }, Sx, n, red);
        // (classes without the LDS copy read the window rows from global memory: four rounds of 8-byte loads in flight)
        const bool w4 = BLOCK > 64 && win && !wl && S.ws == 4 && S.w16;      // (the one-wave classes always hold the LDS copy)
        const uint2* __restrict__ w2 = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(S.win) + (size_t)s0 * 4);
        auto sweep4 = [&](const T* xin, T* out, double shift) {
            for (int p0 = tid; p0 < n; p0 += BLOCK * 4) {
                uint2 wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (p0 + q * BLOCK < n) wv[q] = w2[p0 + q * BLOCK];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = p0 + q * BLOCK;
                    if (p < n) out[p] = (T)sweep_coeff_win4(wv[q], Sx, rs, nlev, lv0[p], (double)xin[p], shift);
                }
            }
        };
        if (w4) sweep4(ms0, key, 1.0);
        else for (int p = tid; p < n; p += BLOCK)
            key[p] = (T)(wl ? sweep_coeff_win(winL + (size_t)p * S.ws, Sx, rs, nlev, lv0[p], (double)ms0[p], 1.0)
                         : win ? sweep_coeff_cached<T>(S, (size_t)s0 + p, Sx, rs, nlev, lv0[p], (double)ms0[p], 1.0)
                               : sweep_coeff<T>(ms0, Sx, rs, nlev, lv0[p], ms0[p], (double)ms0[p], 1.0, strict));
        for (int t = tid; t < ld; t += BLOCK) gvec[t] = (n == 0) ? 0.0 : uvec[t] * lambda;   // :495-498
        if (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the LDS-DMA of stage_rows
        __syncthreads();
        UPROF(1);
        gather_axpy(key, gvec);
        UPROF(2);
        double un2 = 0.0, gn2 = 0.0;
        for (int t = tid; t < ld; t += BLOCK) { un2 += uvec[t] * uvec[t]; gn2 += gvec[t] * gvec[t]; }
        un2 = block_sum<BLOCK>(un2, red);
        gn2 = block_sum<BLOCK>(gn2, red);
        // ---- prev_obj, objective_u_new (pcrpp.cpp:542-573)
        // the user's loss at the gradient point is what the last k_prepare left in objp[u] (same m, same windows):
        // no need to sweep for it again
        const double prev_obj = lambda / 2.0 * un2 + S.objp[u];
        double obj_new = prev_obj, loss_new = 0.0;
        int n_cg = 0, n_ls = 0, ls_free = 0;
        // pcrpp.cpp:787-790; PrimalCR additionally keeps u when no comparable pair exists
        // (cc == 0, pcr.cpp:552)
        const bool skip = (gn2 < 0.0001) || (solver1 && nlev <= 1);
        for (int t = tid; t < ld; t += BLOCK) unew[t] = uvec[t];
        __syncthreads();
        UPROF(6);
        if (!skip) {
            // pcr_tune("ustep_newton"): the exact Newton direction of this user, from the explicit Hessian (k_unewton, pcr_newton.h),
            // replaces the CG below; NaN in its first word = not provided (a user k_unewton does not cover, or a Hessian it could
            // not factor): the CG runs, with the caller's cg_max / cg_tol
            const bool have_dir = dir != nullptr && dir[(size_t)u * ld] == dir[(size_t)u * ld];
            // ---- CG, solve_delta_u_new (pcrpp.cpp:628-647)
            for (int t = tid; t < ld; t += BLOCK) { delta[t] = have_dir ? dir[(size_t)u * ld + t] : 0.0; rr[t] = gvec[t] * -1.0; pv[t] = gvec[t]; }
            const double err = sqrt(gn2) * cg_tol;                              // 0.01 in the reference (:632)
            // The first line-search try needs no pass over the rows: V_I (u - s delta) = m - s sum_k alpha_k (V_I p_k), and
            // b_k = V_I p_k is what every CG iteration computes anyway.  With the window cache on, the gradient point's scores
            // ms0 are not needed again after the gradient sweep, so they carry the running m - s0 sum alpha_k b_k; the sweep
            // writes its coefficients beside b (into the sort's index array, idle until the line search) so that b survives
            // until alpha is known.  Not when the sorted state belongs to a REJECTED V_new (its m is not V_I u, quirk q5), not
            // without the window cache (the sweeps then search ms0), not when T is wider than the index array (fp64 in LDS).
            const bool mrec = !have_dir && !(fault & 4) && win && sizeof(T) <= sizeof(LI);
            T* cst = mrec ? reinterpret_cast<T*>(li) : key;
            ls_free = mrec ? 1 : 0;
            __syncthreads();
            for (int k = 1; k <= (have_dir ? 0 : cg_max); ++k) {                // 10 in the reference (:636)
                for (int t = tid; t < ld; t += BLOCK) { vecT[t] = (T)pv[t]; Hp[t] = pv[t] * lambda; }
                __syncthreads();
                sddmm(key);                                                     // b = V_I p  (:592-594)
                __syncthreads();
                exchange_scores(key, n, r0, r1);
                UPROF(3);
                block_excl_scan<BLOCK>([&](int i) { return (double)key[i]; }, Sx, n, red);
                if (w4) sweep4(key, cst, 0.0);
                else for (int p = tid; p < n; p += BLOCK)
                    cst[p] = (T)(wl ? sweep_coeff_win(winL + (size_t)p * S.ws, Sx, rs, nlev, lv0[p], (double)key[p], 0.0)
                                 : win ? sweep_coeff_cached<T>(S, (size_t)s0 + p, Sx, rs, nlev, lv0[p], (double)key[p], 0.0)
                                       : sweep_coeff<T>(ms0, Sx, rs, nlev, lv0[p], ms0[p], (double)key[p], 0.0, strict));
                __syncthreads();
                UPROF(4);
                gather_axpy(cst, Hp);
                UPROF(5);
                ++n_cg;
                double a = 0.0, b = 0.0;
                for (int t = tid; t < ld; t += BLOCK) { a += pv[t] * Hp[t]; b += rr[t] * pv[t]; }
                const double pHp = block_sum<BLOCK>(a, red);
                const double rp = block_sum<BLOCK>(b, red);
                const double alpha = -1.0 * rp / pHp;
                if (mrec) { const double sa = stepsize0 * alpha; for (int p = tid; p < n; p += BLOCK) ms0[p] = (T)((double)ms0[p] - sa * (double)key[p]); }
                a = 0.0; b = 0.0;
                for (int t = tid; t < ld; t += BLOCK) {
                    delta[t] = delta[t] + pv[t] * alpha;
                    const double rn = rr[t] + Hp[t] * alpha;
                    rr[t] = rn;
                    a += rn * rn;
                    b += rn * Hp[t];
                }
                const double rr2 = block_sum<BLOCK>(a, red);
                const double rHp = block_sum<BLOCK>(b, red);
                UPROF(6);
                if (sqrt(rr2) < err) break;
                const double beta = rHp / pHp;
                for (int t = tid; t < ld; t += BLOCK) pv[t] = rr[t] * -1.0 + pv[t] * beta;
                __syncthreads();
            }
            __syncthreads();
            // ---- line search (pcrpp.cpp:794-813): fresh scores, fresh sort, objective
            double step = stepsize0;
            const int npad = next_pow2(n);
            for (int it = 0; it < 20; ++it) {
                double nn = 0.0;
                for (int t = tid; t < ld; t += BLOCK) {
                    const double v = uvec[t] + delta[t] * -step;
                    unew[t] = v;
                    vecT[t] = (T)v;
                    nn += (double)(T)v * (double)(T)v;
                }
                nn = block_sum<BLOCK>(nn, red);
                __syncthreads();
                UPROF(6);
                if (mrec && it == 0) {                                          // scores of u - s0 delta from the CG's own b_k
                    for (int p = tid; p < n; p += BLOCK) key[p] = ms0[p];
                } else {
                    sddmm(key);                                                 // compute_mm_old (:728-744)
                    if (K > 1) { __syncthreads(); exchange_scores(key, n, r0, r1); }
                }
                UPROF(7);
                for (int p = tid; p < npad; p += BLOCK) {
                    if (p < n) li[p] = LiOps<LI>::pack(lv0[p], (unsigned)p);
                    else { li[p] = LiOps<LI>::pack(0xFFFFu, (unsigned)p); key[p] = (T)0; }
                }
                __syncthreads();
                // update_infor_ui (:684-726): key holds the new scores in the order of the gradient point -- nearly sorted from the
                // third outer iteration on (resort_window), else the full network
                // (tmp lives in the prefix-sum array, the second key array in ms0: the gradient point's scores are not needed again)
                bool resorted = false;
                if constexpr (!BIG) resorted = resort_window<T, LI, BLOCK>(key, li, [&](int p) { return (int)lv0[p]; }, rs, n, reinterpret_cast<int*>(Sx), ms0, S.resort_d, reinterpret_cast<int*>(red));
                if (!resorted) bitonic_sort<T, LI, BLOCK, false, !BIG>(key, li, npad, n);
                UPROF(8);
                // objective_u_new (:542-573) of the tried point.  With the LDS window copy (free again: the CG is over) its windows
                // are searched ONCE, into that copy: the loss reads them (block_objective_win) and the state store below copies
                // them out -- one search pass per rating and level instead of three.  (Not in the one-wave classes: short users, and
                // the extra code costs them a wave per SIMD of registers.)
                if (BLOCK > PCR_WAVE && wl) {
#pragma unroll 1
                    for (int p = tid; p < n; p += BLOCK)
                        find_windows<T, uint16_t>(key, rs, nlev, (int)LiOps<LI>::lev(li[p]), key[p], strict, winL + (size_t)p * S.ws);
                    __syncthreads();
                    loss_new = block_objective_win<T, BLOCK>(key, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, nlev, n, winL, S.ws, Sx, red);
                } else
                    loss_new = block_objective<T, BLOCK>(key, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, nlev, n, Sx, red, strict);
                obj_new = lambda / 2.0 * nn + loss_new;
                ++n_ls;
                UPROF(9);
                if (obj_new < prev_obj) break;
                step /= 2.0;
            }
        }
        __syncthreads();
        // ---- The scores of the last line-search try ARE m = V_I u_new, sorted: leave them as the shard's sorted state
        // (what k_prepare would rebuild from (U_new, V) at the start of the next V step: scores, items, levels, the
        // sorted -> CSR map, the window cache and the loss), so that update_V needs no SDDMM + sort of its own.
        // A skipped user (:787-790) keeps u, so its state stays valid as it is.
        if (!skip && mem == 0) {
            int32_t* stage = reinterpret_cast<int32_t*>(Sx);                  // Sx is free again: (cap + 1) doubles >= n ints
            for (int p = tid; p < n; p += BLOCK) stage[p] = S.sidx[s0 + LiOps<LI>::idx(li[p])];
            __syncthreads();                                                  // all of the old map is read before any of it is rewritten
            for (int p = tid; p < n; p += BLOCK) {
                const LI x = li[p];
                const int lev = (int)LiOps<LI>::lev(x);
                S.ms[s0 + p] = key[p];
                S.slvl[s0 + p] = (uint16_t)lev;
                S.sitem[s0 + p] = itm[LiOps<LI>::idx(x)];
                S.sidx[s0 + p] = stage[p];
                if (S.ws && !(BLOCK > PCR_WAVE && wl)) store_windows<T>(S, (size_t)s0 + p, key, rs, nlev, lev, key[p], strict);
            }
            if (BLOCK > PCR_WAVE && wl) {     // the windows of the last tried point are in the LDS copy (found for its objective)
                if (S.w16) {
                    uint16_t* wg = reinterpret_cast<uint16_t*>(S.win) + (size_t)s0 * S.ws;
#pragma unroll 1
                    for (int i = tid; i < n * S.ws; i += BLOCK) wg[i] = winL[i];
                } else {
                    uint32_t* wg = reinterpret_cast<uint32_t*>(S.win) + (size_t)s0 * S.ws;
#pragma unroll 1
                    for (int i = tid; i < n * S.ws; i += BLOCK) wg[i] = winL[i];
                }
            }
        }
        if (mem == 0) for (int t = tid; t < ld; t += BLOCK) U[(size_t)u * ld + t] = (T)unew[t];
        if (tid == 0 && mem == 0) {
            S.objr[u] = obj_new;
            if (!skip) S.objp[u] = loss_new;
            if (n_cg) atomicAdd(counters + 0, (unsigned long long)n_cg);
            if (n_ls) atomicAdd(counters + 1, (unsigned long long)n_ls);
            // rows of V this user's step gathered: gradient + 2 per CG iteration + 1 per line-search try (diagnostic, only
            // with pcr_tune("count_rows"): a third same-address atomic per user costs the short classes 10-20 %)
            if (fault & 2) {
                const unsigned long long rows = (unsigned long long)n * (unsigned long long)(1 + 2 * n_cg + n_ls - ls_free);
                atomicAdd(counters + 2, rows);
                atomicAdd(cb.rows, rows);
            }
        }
        __syncthreads();
        UPROF(10);
    }
#ifdef PCR_USTEP_PROF
    if (threadIdx.x == 0) {
        const int cls = (K > 1) ? 3 : (BLOCK == 64 ? 0 : BLOCK == 256 ? 1 : 2);
        for (int ph = 0; ph < 11; ++ph) atomicAdd(counters + 4 + cls * 16 + ph, (unsigned long long)prof_[ph]);
        atomicAdd(counters + 4 + cls * 16 + 11, (unsigned long long)(clock64() - tstart_));
        atomicAdd(counters + 4 + cls * 16 + 12, 1ull);
    }
#endif
}

