// pcr_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the PrimalCR /
// PrimalCR++ hot path.  Header-only templates, instantiated in pcr_solver.hip.
//
// Kernel map (reference site -> kernel); design and rooflines: DESIGN.md section 3:
//   K1        comp_m_new, b = u.a in compute_Ha_new        -> k_sddmm     (rating-parallel)
//   K2+K6     get_sorted_mm + objective_new                -> k_prepare   (per user: sort, windows, loss)
//   K4        obtain_g_new  sweep (pcrpp.cpp:214-238)      -> k_vsweep<GRAD> / k_vsweep_wave + k_spmm + k_spmm_fin
//   K5        compute_Ha_new sweep (pcrpp.cpp:294-318)     -> k_vsweep<HV>   / k_vsweep_wave + k_spmm + k_spmm_fin
//   K7        solve_delta_new vector ops (pcrpp.cpp:335)   -> k_cg_init / k_cg_a / k_cg_b / k_cg_c
//   K8        update_u_new (pcrpp.cpp:779-815)             -> k_ustep     (per user, fused; K workgroups per long user)
//   K10       compute_pairwise_error_ndcg (util.cpp:434)   -> k_eval2 (sorted) / k_eval (brute force)
//             pmf-predict.cpp:56-64                        -> k_predict
//
// Formulation (replaces the sequential two-pointer sweep with data-parallel primitives,
// same result): a user's ratings are sorted by (level, m), so every rating level is one
// sorted run.  For item p of level l and another level l':
//     l' > l : partners are the run prefix {q : m_q <= m_p + 1}   (pcrpp.cpp:218)
//     l' < l : partners are the run suffix {q : m_q >= m_p - 1}   (pcrpp.cpp:224)
// found by binary search ONCE per sorted state (window cache); counts come from the indices, sums
// from one fp64 exclusive prefix sum over the sorted order.  Every later sweep is O(len * T).
//
// Layout: factor rows are padded to ld = roundup(k, 4) elements so every row is 16-byte
// aligned; a row is read by a group of G lanes (G = pow2 >= ld/VEC, <= 64), one 16-byte
// vector per lane -> each wave-instruction moves 64/G whole rows, coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PCR_WAVE 64

template <typename T> struct VecOf;
template <> struct VecOf<float>  { typedef float4 type;  static constexpr int N = 4; };
template <> struct VecOf<double> { typedef double2 type; static constexpr int N = 2; };

__device__ __forceinline__ float  vdot(const float4& a, const float4& b)  { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ double vdot(const double2& a, const double2& b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float  velem(const float4& a, int e)  { return e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w; }
__device__ __forceinline__ double velem(const double2& a, int e) { return e == 0 ? a.x : a.y; }

struct Geo { int r, ld, nchunk, G; };   // rank, padded row length, 16-byte chunks per row, lanes per row
// element offset of row `row` of a factor matrix: ONE v_mad_u64_u32.  (size_t)row * geo.ld with the two ints sign-extended is
// a full 64 x 64-bit multiply -- seven VALU instructions per gathered row in kernels that sit at the issue limit.
// Row indices are never negative and ld > 0.
__device__ __forceinline__ size_t row_off(int row, const Geo& geo) { return (size_t)((unsigned long long)(unsigned)row * (unsigned)geo.ld); }

// this rank's training shard on the device
template <typename T>
struct Shard {
    int64_t nu, nnz;
    int d2;
    const int64_t* uptr;       // nu+1   user -> CSR offset
    const int32_t* item;       // nnz    CSR order
    const uint16_t* lvl;       // nnz    CSR order, dense level inside the user
    const int64_t* runofs;     // nu+1
    const int32_t* runstart;   // per user T_u+1 cumulative level counts
    // (level, m)-sorted state written by k_prepare
    T* ms;                     // nnz
    int32_t* sitem;            // nnz    item id at sorted position
    uint16_t* slvl;            // nnz
    int32_t* sidx;             // nnz    index INSIDE THE USER'S CSR SEGMENT of the rating at each sorted position: the only map
                               //        between the two orders.  Per-rating values that cross kernels (the SDDMM's b, the
                               //        sweeps' c) live in CSR order; a sweep reads b and writes c through sidx -- a
                               //        permutation inside its own user's segment, whole cache lines -- and the item-major
                               //        k_spmm reads c through a STATIC CSC -> CSR map.  (Round 1 kept a dynamic CSC <-> sorted
                               //        map instead, rewritten by every sort: a 4-byte scatter per rating into 64-byte lines
                               //        spread over the whole shard -- 3.6x the algorithmic write traffic in k_prepare.)
    double* objp;              // nu     per-user loss partial (no regulariser)
    double* objr;              // nu     k_ustep: obj_u of the returned u (loss + lambda/2 |u|^2, pcrpp.cpp:835)
    // window cache: for sorted position p and every OTHER level l' (slot = l' < l ? l' : l'-1) the
    // boundary index of the active prefix / suffix of run l'.  Depends on m only, so k_prepare
    // finds it once and every sweep of the V step (gradient + <=10 Hessian-vector products) and of
    // the U step (gradient, objective, CG) reuses it.  ws = slots per item (0 = cache disabled:
    // too many levels; the sweeps then search).
    // Entries are positions inside the user: 16 bits wide when no user of the shard has 65536 ratings or more (w16),
    // else 32 -- half the bytes of what is the largest per-rating array of the state (4 slots x 5 levels).
    void* win;                 // nnz * ws entries of uint16_t (w16) or uint32_t
    int ws, w16;
    // nearly-sorted fast path of the sorts (resort_window): half-width of the window a rating may have moved by (0 = always
    // the full bitonic network); prev_valid: sidx / slvl hold a valid permutation of every user (any earlier sorted state)
    int resort_d, prev_valid;
};

// ---------------------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
// ---------------------------------------------------------------------------------------

// lane ^ 1 and lane ^ 2 exchanges as DPP quad permutes (a VALU modifier: no LDS-pipe ds_bpermute, no address arithmetic)
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }    // quad_perm [2,3,0,1]
// lane ^ 4, ^ 8: ds_swizzle in bit mode (LDS crossbar, but no address VGPR and no address arithmetic);
// lane ^ 16, ^ 32: gfx950's v_permlane16_swap / v_permlane32_swap (VALU) + a select.  (tools/ubench/xor_probe.hip)
__device__ __forceinline__ int swz_xor4(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x101F); }
__device__ __forceinline__ int swz_xor8(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x201F); }
__device__ __forceinline__ int perm_xor16(int v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ int perm_xor32(int v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
}
template <int OFF> __device__ __forceinline__ int lane_xor_i(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "power of two below 64");
    return OFF == 1 ? dpp_xor1(v) : OFF == 2 ? dpp_xor2(v) : OFF == 4 ? swz_xor4(v) : OFF == 8 ? swz_xor8(v) : OFF == 16 ? perm_xor16(v) : perm_xor32(v);
}
template <int OFF> __device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor_i<OFF>(__float_as_int(v))); }
template <int OFF> __device__ __forceinline__ double lane_xor(double v) {
    return __hiloint2double(lane_xor_i<OFF>(__double2hiint(v)), lane_xor_i<OFF>(__double2loint(v)));
}
__device__ __forceinline__ float lane_xor1(float v) { return __int_as_float(dpp_xor1(__float_as_int(v))); }
__device__ __forceinline__ float lane_xor2(float v) { return __int_as_float(dpp_xor2(__float_as_int(v))); }
__device__ __forceinline__ double lane_xor1(double v) {
    return __hiloint2double(dpp_xor1(__double2hiint(v)), dpp_xor1(__double2loint(v)));
}
__device__ __forceinline__ double lane_xor2(double v) {
    return __hiloint2double(dpp_xor2(__double2hiint(v)), dpp_xor2(__double2loint(v)));
}

// wave-wide inclusive scan with DPP only: Kogge-Stone inside each 16-lane row (row_shr 1, 2, 4, 8, zero fill), then lane 15
// of rows 0 / 2 added to rows 1 / 3 (row_bcast15) and lane 31 to rows 2, 3 (row_bcast31): 6 VALU steps, no LDS crossbar
// (tools/ubench/scan_probe.hip)
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_zero_fill(double v) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_zero_fill<0x111, 0xF>(v);
    v += dpp_zero_fill<0x112, 0xF>(v);
    v += dpp_zero_fill<0x114, 0xF>(v);
    v += dpp_zero_fill<0x118, 0xF>(v);
    v += dpp_zero_fill<0x142, 0xA>(v);
    v += dpp_zero_fill<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ double lane63(double v) {            // the last lane's value, wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ double wave_sum(double v) {
    v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}

// ordering point for LDS traffic that stays inside one wave (the LDS serves a wave's accesses in program order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The per-user primitives below are written for a "team" of BLOCK threads.  BLOCK = 64 is one wave: its ordering points are
// wave-level and its thread index is the lane, so a one-wave team runs unchanged as one of the eight independent waves of
// a 512-thread workgroup (k_prepare_all, k_vsweep_all) as well as in a 64-thread workgroup of its own.
template <int BLOCK> __device__ __forceinline__ void bsync() { if (BLOCK == PCR_WAVE) wave_sync(); else __syncthreads(); }
template <int BLOCK> __device__ __forceinline__ int btid() { return BLOCK == PCR_WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }

// total to every thread; red: LDS, >= BLOCK/64 doubles
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    if (BLOCK == PCR_WAVE) return v;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / PCR_WAVE; ++w) t += red[w];
    return t;
}

// out[i] = sum_{q<i} f(q) for i in [0, n]; fp64; strided rounds keep LDS access conflict-free
template <int BLOCK, class F>
__device__ __forceinline__ void block_excl_scan(F f, double* out, int n, double* red) {
    const int tid = btid<BLOCK>(), lane = tid & 63, wid = tid >> 6;
    double carry = 0.0;
    for (int base = 0; base < n; base += BLOCK) {
        const int i = base + tid;
        const double v = (i < n) ? f(i) : 0.0;
        const double inc = wave_incl_scan(v);
        double woff = 0.0, total;
        if (BLOCK > PCR_WAVE) {
            __syncthreads();
            if (lane == 63) red[wid] = inc;
            __syncthreads();
            total = 0.0;
#pragma unroll
            for (int w = 0; w < BLOCK / PCR_WAVE; ++w) {
                double x = red[w];
                if (w < wid) woff += x;
                total += x;
            }
        } else {
            total = lane63(inc);
        }
        if (i < n) out[i] = carry + woff + inc - v;
        carry += total;
    }
    if (tid == 0) out[n] = carry;
    bsync<BLOCK>();
}

// packed (level, index): LDS-resident users use 32 bits (level<<16 | idx), users that live in
// global scratch use 64 bits (level<<32 | idx)
template <typename LI> struct LiOps;
template <> struct LiOps<uint32_t> {
    static constexpr int SH = 16;
    static __device__ __forceinline__ uint32_t pack(unsigned lv, unsigned idx) { return (lv << 16) | idx; }
    static __device__ __forceinline__ unsigned lev(uint32_t x) { return x >> 16; }
    static __device__ __forceinline__ unsigned idx(uint32_t x) { return x & 0xFFFFu; }
};
template <> struct LiOps<uint64_t> {
    static constexpr int SH = 32;
    static __device__ __forceinline__ uint64_t pack(unsigned lv, unsigned idx) { return ((uint64_t)lv << 32) | idx; }
    static __device__ __forceinline__ unsigned lev(uint64_t x) { return (unsigned)(x >> 32); }
    static __device__ __forceinline__ unsigned idx(uint64_t x) { return (unsigned)(x & 0xFFFFFFFFu); }
};

// ascending bitonic sort of (key, li) by (level, key); npad = pow2 >= n, padding carries the
// maximum level so it sinks to the end.  Tie order among equal (level, key) is irrelevant to
// every sum computed from the order (the reference's std::sort is unstable too).
// TIE = true additionally orders equal (level, key) by DESCENDING index (k_eval2: the lowest index then
// sits at the end of its run and is taken first).
// Compare-exchange network over LDS.  With stride j <= 64 the pairs a wave works on (64 consecutive t) lie in ITS OWN
// aligned 128-element chunk, in every such stage alike, so between two short-stride stages a wave-level ordering point
// replaces the workgroup barrier: of the 78 stages of a 4096-element sort only 21 need __syncthreads().
// (INLDS = false: the arrays live in global scratch, every stage keeps the workgroup barrier.)
template <typename T, typename LI, int BLOCK, bool TIE = false, bool INLDS = true>
__device__ __forceinline__ void bitonic_sort(T* key, LI* li, int npad) {
    const int tid = btid<BLOCK>();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (npad >> 1); t += BLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                T ka = key[i], kb = key[l];
                LI la = li[i], lb = li[l];
                unsigned va = LiOps<LI>::lev(la), vb = LiOps<LI>::lev(lb);
                bool b_lt_a = (vb < va) || (vb == va && kb < ka);
                bool a_lt_b = (va < vb) || (va == vb && ka < kb);
                if (TIE && va == vb && ka == kb) {
                    const unsigned ia = LiOps<LI>::idx(la), ib = LiOps<LI>::idx(lb);
                    b_lt_a = ib > ia; a_lt_b = ia > ib;
                }
                bool sw = up ? b_lt_a : a_lt_b;
                if (sw) { key[i] = kb; key[l] = ka; li[i] = lb; li[l] = la; }
            }
            const int jnext = j > 1 ? (j >> 1) : k;                 // stride of the next stage (first stage of the next phase: k)
            const bool last = (j == 1 && k == npad);
            if (INLDS && !last && j <= 64 && jnext <= 64) wave_sync(); else bsync<BLOCK>();
        }
    }
}

// Nearly-sorted fast path.  From the third outer iteration on, a user's (level, m) order barely moves between two sorted
// states (tools/exp_resort.py, ml1m shape: the largest displacement of any rating of a user is <= 8 positions for 70 % of the
// rating mass at iteration 5, 90 % at 14, 98 % at 25 -- while the full bitonic network costs 45-78 stages of LDS round trip +
// ordering point whatever the input).  key[0, n) / li[0, n) hold the NEW scores in the PREVIOUS order (levels are static, so
// every level is already one contiguous run): the new rank of position p follows from the inversions inside a window of +-D
// positions of its run,   rank = p - #{q in [p - D, p): key_q > key_p} + #{q in (p, p + D]: key_q < key_p}   (stable),
// a scatter of p to tmp[rank], and a gather through a second array: 2 D + 14 LDS operations per rating and 7 ordering points.
// The windowed count is exact only if nothing moved further than D; that is VERIFIED, not assumed: the result is accepted
// only if every slot of tmp was filled (n writes into n slots: a bijection) and the permuted keys ascend inside every run --
// then it IS a (level, m)-sorted order, and any such order gives the same sums (tie order is irrelevant, see bitonic_sort).
// Otherwise key / li are untouched and the caller runs the bitonic network.  tmp: n ints, key2: n T's (both LDS, distinct from
// key / li; key2 == nullptr: no room, no fast path).  No per-thread arrays: the kernels around it are register-bound.
#ifdef PCR_RESORT_STAT
__device__ unsigned long long g_resort_stat[4];      // users (fast path taken, fallen back), their ratings
#endif
template <typename T, typename LI, int BLOCK, class LevF>
__device__ __forceinline__ bool resort_window_d(T* key, LI* li, LevF levf, const int* rs, int n, int* tmp, T* key2, int D, int* flag) {
    static_assert(sizeof(LI) <= sizeof(T) || sizeof(LI) == 4, "li is permuted through key2's bytes");
    const int tid = btid<BLOCK>();
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) tmp[p] = -1;
    if (BLOCK > PCR_WAVE && tid == 0) *flag = 0;               // (flag: one LDS word for the team's verdict; __syncthreads_or would
    bsync<BLOCK>();                                             // add static LDS to kernels that ask for all 160 KB dynamically)
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) {
        const int lev = levf(p);
        const int lo = max(rs[lev], p - D), hi = min(rs[lev + 1], p + D + 1);
        const T kp = key[p];
        int r = p;
#pragma unroll 2
        for (int q = lo; q < p; ++q) r -= (key[q] > kp) ? 1 : 0;
#pragma unroll 2
        for (int q = p + 1; q < hi; ++q) r += (key[q] < kp) ? 1 : 0;
        tmp[r] = p;
    }
    bsync<BLOCK>();
    int bad = 0;
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) {
        const int src = tmp[p];
        bad |= (src < 0) ? 1 : 0;
        key2[p] = key[src < 0 ? 0 : src];
    }
    bsync<BLOCK>();
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK)
        if (p + 1 < rs[levf(p) + 1]) bad |= (key2[p + 1] < key2[p]) ? 1 : 0;          // the next position belongs to the same run
    if (BLOCK == PCR_WAVE) bad = __any(bad);
    else { if (bad) *flag = 1; __syncthreads(); bad = *flag; }
#ifdef PCR_RESORT_STAT
    if (tid == 0) { atomicAdd(&g_resort_stat[bad ? 1 : 0], 1ull); atomicAdd(&g_resort_stat[bad ? 3 : 2], (unsigned long long)n); }
#endif
    if (bad) return false;
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) key[p] = key2[p];
    bsync<BLOCK>();
    LI* li2 = reinterpret_cast<LI*>(key2);
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) li2[p] = li[tmp[p]];
    bsync<BLOCK>();
#pragma unroll 1
    for (int p = tid; p < n; p += BLOCK) li[p] = li2[p];
    bsync<BLOCK>();
    return true;
}

// two tiers: a narrow window first (most users, most iterations), four times as wide for the users that fail it; then the network
template <typename T, typename LI, int BLOCK, class LevF>
__device__ __forceinline__ bool resort_window(T* key, LI* li, LevF levf, const int* rs, int n, int* tmp, T* key2, int D, int* flag) {
    if (D <= 0 || !key2) return false;
#pragma unroll 1
    for (int tier = 0; tier < 2; ++tier) {                     // (one copy of the body: the kernels around it are register-bound)
        if (tier) { bsync<BLOCK>(); D *= 4; if (D >= n) break; }   // (the verdict word and tmp are reused)
        if (resort_window_d<T, LI, BLOCK>(key, li, levf, rs, n, tmp, key2, D, flag)) return true;
    }
    return false;
}

// first index in [s,e) with a[q] > x   (= s + #{a[q] <= x})
template <typename T>
__device__ __forceinline__ int ubound(const T* a, int s, int e, T x) {
    while (s < e) { int m = (s + e) >> 1; if (a[m] <= x) s = m + 1; else e = m; }
    return s;
}
// first index in [s,e) with a[q] >= x  (= s + #{a[q] < x})
template <typename T>
__device__ __forceinline__ int lbound(const T* a, int s, int e, T x) {
    while (s < e) { int m = (s + e) >> 1; if (a[m] < x) s = m + 1; else e = m; }
    return s;
}

// Sweep coefficient of one item (pcrpp.cpp:230-238 with x = m, shift = 1; :310-318 with x = b,
// shift = 0).  ms: (level, m)-sorted scores, S: exclusive prefix sum of x over that order,
// rs: run boundaries.  strict = PrimalCR's `mask < 1.0` (pcr.cpp:137) instead of the inclusive
// windows of PrimalCR++ (pcrpp.cpp:218,224).
template <typename T>
__device__ __forceinline__ double sweep_coeff(const T* ms, const double* S, const int* rs, int nlev, int lev,
                                              T mp, double xp, double shift, int strict) {
    double acc = 0.0;
    const T lo = mp - (T)1, hi = mp + (T)1;
    for (int l = 0; l < nlev; ++l) {
        if (l == lev) continue;
        const int s = rs[l], e = rs[l + 1];
        if (l < lev) {
            const int w = strict ? ubound(ms, s, e, lo) : lbound(ms, s, e, lo);
            acc += (double)(e - w) * (xp - shift) - (S[e] - S[w]);
        } else {
            const int w = strict ? lbound(ms, s, e, hi) : ubound(ms, s, e, hi);
            acc += (double)(w - s) * (xp + shift) - (S[w] - S[s]);
        }
    }
    return 2.0 * acc;
}

// boundaries of item (lev, mp) in every other run -> w[slot]
template <typename T, typename W>
__device__ __forceinline__ void find_windows(const T* ms, const int* rs, int nlev, int lev, T mp, int strict, W* w) {
    const T lo = mp - (T)1, hi = mp + (T)1;
    for (int l = 0; l < nlev; ++l) {
        if (l == lev) continue;
        const int s = rs[l], e = rs[l + 1];
        if (l < lev) w[l] = (W)(strict ? ubound(ms, s, e, lo) : lbound(ms, s, e, lo));
        else w[l - 1] = (W)(strict ? lbound(ms, s, e, hi) : ubound(ms, s, e, hi));
    }
}
// the window row of rating `row` (a position in the shard's sorted state), whatever the entry width
template <typename T>
__device__ __forceinline__ void store_windows(const Shard<T>& S, size_t row, const T* ms, const int* rs, int nlev, int lev, T mp, int strict) {
    if (S.w16) find_windows<T>(ms, rs, nlev, lev, mp, strict, reinterpret_cast<uint16_t*>(S.win) + row * S.ws);
    else find_windows<T>(ms, rs, nlev, lev, mp, strict, reinterpret_cast<uint32_t*>(S.win) + row * S.ws);
}

// sweep_coeff with cached boundaries (w: ws slots of this item)
template <typename W>
__device__ __forceinline__ double sweep_coeff_win(const W* __restrict__ w, const double* S, const int* rs, int nlev,
                                                  int lev, double xp, double shift) {
    double acc = 0.0;
    for (int l = 0; l < lev; ++l) {
        const int wi = (int)w[l], e = rs[l + 1];
        acc += (double)(e - wi) * (xp - shift) - (S[e] - S[wi]);
    }
    for (int l = lev + 1; l < nlev; ++l) {
        const int wi = (int)w[l - 1], s0 = rs[l];
        acc += (double)(wi - s0) * (xp + shift) - (S[wi] - S[s0]);
    }
    return 2.0 * acc;
}

// sweep_coeff_win for the common layout -- at most 5 levels (ws == 4), 16-bit entries: the rating's four boundaries arrive as
// ONE 8-byte load (wv), the loop over the other levels is unrolled with the slot chosen by a select.  Same terms in the same
// order as sweep_coeff_win (levels ascending), so the result is bitwise the same.
__device__ __forceinline__ double sweep_coeff_win4(uint2 wv, const double* S, const int* rs, int nlev, int lev, double xp, double shift) {
    const int w[4] = {(int)(wv.x & 0xFFFFu), (int)(wv.x >> 16), (int)(wv.y & 0xFFFFu), (int)(wv.y >> 16)};
    double acc = 0.0;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        if (l >= nlev || l == lev) continue;
        if (l < lev) {
            const int wi = w[l < 4 ? l : 3], e = rs[l + 1];
            acc += (double)(e - wi) * (xp - shift) - (S[e] - S[wi]);
        } else {
            const int wi = w[l - 1 >= 0 ? l - 1 : 0], s0 = rs[l];
            acc += (double)(wi - s0) * (xp + shift) - (S[wi] - S[s0]);
        }
    }
    return 2.0 * acc;
}
// the sweep's output loop for that layout: the per-rating loads (level, boundaries, CSR index) of FOUR rounds are issued before
// the first is used.  The sweep is bound by bytes in flight on large shards (Little's law: ~35 % occupancy x 3 small loads per
// wave = 1.5 TB/s on the Netflix shape); on ml1m the launch is as long as its longest user and this changes nothing.
template <typename T, int STRIDE, bool HV>
__device__ __forceinline__ void sweep_out4(const Shard<T>& S, int64_t s0, int n, int nlev, int tid, const T* xs, const double* Sx,
                                           const int* rs, T* __restrict__ c_out) {
    const uint2* __restrict__ w2 = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(S.win) + (size_t)s0 * 4);
    for (int p0 = tid; p0 < n; p0 += STRIDE * 4) {
        uint2 wv[4];
        int lv[4], si[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * STRIDE;
            if (p < n) { wv[q] = w2[p]; lv[q] = S.slvl[s0 + p]; si[q] = S.sidx[s0 + p]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + q * STRIDE;
            if (p < n) c_out[s0 + si[q]] = (T)sweep_coeff_win4(wv[q], Sx, rs, nlev, lv[q], (double)xs[p], HV ? 0.0 : 1.0);
        }
    }
}

// sweep_coeff_win on the shard's window cache, whatever the entry width
template <typename T>
__device__ __forceinline__ double sweep_coeff_cached(const Shard<T>& Sh, size_t row, const double* S, const int* rs, int nlev,
                                                     int lev, double xp, double shift) {
    return Sh.w16 ? sweep_coeff_win(reinterpret_cast<const uint16_t*>(Sh.win) + row * Sh.ws, S, rs, nlev, lev, xp, shift)
                  : sweep_coeff_win(reinterpret_cast<const uint32_t*>(Sh.win) + row * Sh.ws, S, rs, nlev, lev, xp, shift);
}

// block_objective with cached boundaries (win: the user's window rows, ws slots each)
template <typename T, int BLOCK, class LevF, typename W>
__device__ __forceinline__ double block_objective_win(const T* ms, LevF levf, const int* rs, int nlev, int n,
                                                      const W* __restrict__ win, int ws, double* S, double* red) {
    const int tid = btid<BLOCK>();
    double part = 0.0;
    block_excl_scan<BLOCK>([&](int i) { return (double)ms[i] - 1.0; }, S, n, red);
    for (int p = tid; p < n; p += BLOCK) {
        const int lev = levf(p);
        const double m = (double)ms[p];
        const W* w = win + (size_t)p * ws;
        for (int l = lev + 1; l < nlev; ++l) {
            const int s0 = rs[l], wi = (int)w[l - 1];
            part += (double)(wi - s0) * m * m - 2.0 * m * (S[wi] - S[s0]);
        }
    }
    bsync<BLOCK>();
    block_excl_scan<BLOCK>([&](int i) { double d = (double)ms[i] - 1.0; return d * d; }, S, n, red);
    for (int p = tid; p < n; p += BLOCK) {
        const int lev = levf(p);
        const W* w = win + (size_t)p * ws;
        for (int l = lev + 1; l < nlev; ++l) part += S[w[l - 1]] - S[rs[l]];
    }
    bsync<BLOCK>();
    return block_sum<BLOCK>(part, red);
}

// Loss of one user (pcrpp.cpp:392-407): sum over items p and higher levels l' of
//   cnt*m_p^2 - 2 m_p * sum(m_q - 1) + sum((m_q - 1)^2)  over the active prefix of run l'.
// Two passes share ONE fp64 prefix array S (LDS budget); levf(p) = level of sorted position p.
template <typename T, int BLOCK, class LevF>
__device__ __forceinline__ double block_objective(const T* ms, LevF levf, const int* rs, int nlev, int n,
                                                  double* S, double* red, int strict) {
    const int tid = btid<BLOCK>();
    double part = 0.0;
    block_excl_scan<BLOCK>([&](int i) { return (double)ms[i] - 1.0; }, S, n, red);
    for (int p = tid; p < n; p += BLOCK) {
        const int lev = levf(p);
        const T mp = ms[p];
        const T hi = mp + (T)1;
        const double m = (double)mp;
        for (int l = lev + 1; l < nlev; ++l) {
            const int s = rs[l], e = rs[l + 1];
            const int w = strict ? lbound(ms, s, e, hi) : ubound(ms, s, e, hi);
            part += (double)(w - s) * m * m - 2.0 * m * (S[w] - S[s]);
        }
    }
    bsync<BLOCK>();
    block_excl_scan<BLOCK>([&](int i) { double d = (double)ms[i] - 1.0; return d * d; }, S, n, red);
    for (int p = tid; p < n; p += BLOCK) {
        const int lev = levf(p);
        const T hi = ms[p] + (T)1;
        for (int l = lev + 1; l < nlev; ++l) {
            const int s = rs[l], e = rs[l + 1];
            const int w = strict ? lbound(ms, s, e, hi) : ubound(ms, s, e, hi);
            part += S[w] - S[s];
        }
    }
    bsync<BLOCK>();
    return block_sum<BLOCK>(part, red);
}

// out[p] = vec . M[rows[p]]  for p in [0, n)   (SDDMM of one user; pcrpp.cpp:28-31, :266-271,
// :592-594, :735-742).  vecT: LDS, ld entries of T.  rows: item ids, staged in LDS by the caller
// (no dependent global index load in front of the row load).  G lanes per row; PCR_UNR rows are
// in flight per lane group (memory-level parallelism: the gathers are latency-bound).  Rows longer
// than G chunks are handled by an outer pass per chunk set.
#define PCR_UNR 8

// Sum 8 per-lane values over the G lanes of each lane group (G = 8, 16, 32 or 64) with 9-10
// constant-offset shuffles instead of 8 * log2(G): at xor 1, 2, 4 each lane keeps half of its
// values and sends the other half, so after three steps it owns ONE row's partial; the remaining
// steps are plain butterflies.  Returns the total of row rho(g) = 4*(g&1) + (g&2) + ((g>>2)&1),
// identical in the G/8 lanes that share g&7.
template <typename T>
__device__ __forceinline__ T group_reduce8(const T (&a)[8], int g, int G) {
    T b[4], c[2], d;
    const bool b0 = g & 1, b1 = g & 2, b2 = g & 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const T send = b0 ? a[i] : a[i + 4], keep = b0 ? a[i + 4] : a[i];
        b[i] = keep + lane_xor1(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const T send = b1 ? b[i] : b[i + 2], keep = b1 ? b[i + 2] : b[i];
        c[i] = keep + lane_xor2(send);
    }
    {
        const T send = b2 ? c[0] : c[1], keep = b2 ? c[1] : c[0];
        d = keep + lane_xor<4>(send);
    }
    if (G > 8) d += lane_xor<8>(d);
    if (G > 16) d += lane_xor<16>(d);
    if (G > 32) d += lane_xor<32>(d);
    return d;
}

// same for 4 values (G = 4 .. 64): returns the total of row rho4(g) = 2*(g&1) + ((g>>1)&1)
template <typename T>
__device__ __forceinline__ T group_reduce4(const T (&a)[4], int g, int G) {
    T b[2], d;
    const bool b0 = g & 1, b1 = g & 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const T send = b0 ? a[i] : a[i + 2], keep = b0 ? a[i + 2] : a[i];
        b[i] = keep + lane_xor1(send);
    }
    {
        const T send = b1 ? b[0] : b[1], keep = b1 ? b[1] : b[0];
        d = keep + lane_xor2(send);
    }
    if (G > 4) d += lane_xor<4>(d);
    if (G > 8) d += lane_xor<8>(d);
    if (G > 16) d += lane_xor<16>(d);
    if (G > 32) d += lane_xor<32>(d);
    return d;
}

// The per-user (workgroup) primitives keep 4 rows in flight per lane group: k_ustep is register-bound
// (occupancy), and its many resident waves provide the memory-level parallelism instead.
// rows in flight per lane group of the per-user primitives: 4 in the one-wave / 256-thread kernels (register-bound:
// occupancy provides the memory-level parallelism), 8 in the 512-thread kernels (one workgroup per CU anyway)
#ifndef PCR_BUNR
#define PCR_BUNR (BLOCK >= 512 ? 8 : 4)
#endif
// LROWS: M is the workgroup's LDS image of rows [r0, n) (stage_rows), lstride elements per row.
#define PCR_LDS __attribute__((address_space(3)))
// one 16-byte ds_read_b128 from the workgroup's LDS (p: generic pointer known to point into LDS)
__device__ __forceinline__ float4 lds_load_vec(const float* p) {
    typedef float nat __attribute__((ext_vector_type(4)));
    const nat v = *(const PCR_LDS nat*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 lds_load_vec(const double* p) {
    typedef double nat __attribute__((ext_vector_type(2)));
    const nat v = *(const PCR_LDS nat*)p;
    return make_double2(v.x, v.y);
}
template <typename T, int BLOCK, bool LROWS = false, int UNR = PCR_BUNR>
__device__ __forceinline__ void block_sddmm(const T* __restrict__ M, const T* vecT, const int32_t* rows, int n,
                                            T* out, const Geo& geo, int r0 = 0, int lstride = 0) {      // rows [r0, n)
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    const int G = geo.G, g = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = BLOCK / G;
    static_assert(UNR == 4 || UNR == 8, "rows in flight per lane group");
    const int rho = (UNR == 8) ? 4 * (g & 1) + (g & 2) + ((g >> 2) & 1) : 2 * (g & 1) + ((g >> 1) & 1);
    for (int k = 0; k * G < geo.nchunk; ++k) {
        const int ch = g + k * G;
        const bool act = ch < geo.nchunk;
        V uv;
        if (act) uv = *reinterpret_cast<const V*>(vecT + ch * VEC);
        for (int base = r0 + grp; base < n; base += ngrp * UNR) {
            V rv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int row = base + q * ngrp;
                if (row < n && act) {
                    if (LROWS) rv[q] = lds_load_vec(M + (row - r0) * lstride + ch * VEC);
                    else rv[q] = *reinterpret_cast<const V*>(M + row_off(rows[row], geo) + ch * VEC);
                }
            }
            T part[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) part[q] = (act && base + q * ngrp < n) ? vdot(rv[q], uv) : (T)0;
            if (G >= UNR) {
                T tot;                                               // whole lane groups are active here
                if constexpr (UNR == 8) tot = group_reduce8<T>(part, g, G); else tot = group_reduce4<T>(part, g, G);
                const int row = base + rho * ngrp;
                if (g < UNR && row < n) out[row] = (k == 0) ? tot : out[row] + tot;
            } else {
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    T v = part[q];
                    if (G > 2) v += __shfl_xor(v, 2);
                    if (G > 1) v += __shfl_xor(v, 1);
                    const int row = base + q * ngrp;
                    if (g == 0 && row < n) out[row] = (k == 0) ? v : out[row] + v;
                }
            }
        }
    }
}

// outvec[0..ld) += sum_{p in [r0,n)} c[p] * M[rows[p]]   (pcrpp.cpp:536, :622).  fp64 accumulation.
// wbuf: LDS, (BLOCK/64) * ld doubles.  Ends with a barrier; outvec valid for all threads.
// assign = true: outvec = sum (a partial, for the multi-workgroup exchange) instead of +=.
template <typename T, typename CT, int BLOCK, bool LROWS = false, int UNR = PCR_BUNR>
__device__ __forceinline__ void block_gather_axpy(const T* __restrict__ M, const int32_t* rows, const CT* c, int n,
                                                  double* outvec, double* wbuf, const Geo& geo, int r0 = 0, bool assign = false,
                                                  int lstride = 0) {
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    const int G = geo.G, g = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = BLOCK / G;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k * G < geo.nchunk; ++k) {
        const int ch = g + k * G;
        const bool act = ch < geo.nchunk;
        // A lane group's running sum (its 1/ngrp share of the user's rows, a few hundred terms at most) is kept in T, as the
        // running row of k_spmm is; the sums across groups and workgroups are fp64.  For T = float the fp64 multiply-adds
        // and conversions were most of this loop's VALU work and four more registers per lane.
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = (T)0;
        for (int base = r0 + grp; base < n; base += ngrp * UNR) {
            V rv[UNR];
            T cc[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int row = base + q * ngrp;
                cc[q] = (T)0;
                if (row < n && act) {
                    cc[q] = (T)c[row];
                    if (LROWS) rv[q] = lds_load_vec(M + (row - r0) * lstride + ch * VEC);
                    else rv[q] = *reinterpret_cast<const V*>(M + row_off(rows[row], geo) + ch * VEC);
                }
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int row = base + q * ngrp;
                if (row < n && act) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += cc[q] * velem(rv[q], e);
                }
            }
        }
        // groups of one wave -> one vector
        double accd[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) accd[e] = (double)acc[e];
        for (int off = G; off < PCR_WAVE; off <<= 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) accd[e] += __shfl_xor(accd[e], off);
        }
        if (k == 0) __syncthreads();
        if (lane < G && act) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) wbuf[wid * geo.ld + ch * VEC + e] = accd[e];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < geo.ld; t += BLOCK) {
        double sum = 0.0;
        for (int w = 0; w < BLOCK / PCR_WAVE; ++w) sum += wbuf[w * geo.ld + t];
        outvec[t] = assign ? sum : outvec[t] + sum;
    }
    __syncthreads();
}

// LDS image of rows [q0, q1) of one user: img[(row - q0) * nchp + ch] (16-byte chunks; nchp = chunks per LDS row, odd so
// that the 16 lanes of a ds_read_b128 quarter-wave that read the same chunk of consecutive rows hit distinct banks).
// Filled by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, so a wave keeps dozens of row pieces in flight):
// one wave-instruction writes 64 consecutive chunks = wave-uniform base + lane * 16 B, the source address is per lane;
// lanes that fall on a pad chunk or past the end are masked off.  The caller waits (vmcnt(0) + barrier) before reading.
template <typename T, int BLOCK>
__device__ __forceinline__ void stage_rows(const T* __restrict__ M, const int32_t* rows, int q0, int q1, T* img,
                                           const Geo& geo, int nchp) {
    constexpr int VEC = VecOf<T>::N;
    const int total = (q1 - q0) * nchp;
    const int lane = threadIdx.x & 63;
    int row = (int)threadIdx.x / nchp, col = (int)threadIdx.x - row * nchp;
    const int drow = BLOCK / nchp, dcol = BLOCK - drow * nchp;
    for (int base = (int)(threadIdx.x & ~63u); base < total; base += BLOCK) {
        const int ubase = __builtin_amdgcn_readfirstlane(base);
        if (ubase + lane < total && col < geo.nchunk) {
            const T* src = M + row_off(rows[q0 + row], geo) + col * VEC;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (PCR_LDS void*)((PCR_LDS char*)img + (size_t)ubase * 16), 16, 0, 0);
        }
        row += drow; col += dcol;
        if (col >= nchp) { col -= nchp; ++row; }
    }
}

// carve typed arrays out of a byte region (16-byte aligned pieces)
struct Carver {
    char* p;
    __device__ explicit Carver(char* base) : p(base) {}
    template <class X> __device__ X* take(size_t n) {
        X* r = reinterpret_cast<X*>(p);
        p += (n * sizeof(X) + 15) & ~(size_t)15;
        return r;
    }
};
static inline size_t carve_bytes(size_t n, size_t elt) { return (n * elt + 15) & ~(size_t)15; }

__device__ __forceinline__ int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// ---------------------------------------------------------------------------------------
// k_sddmm: out[z] = U[ruser[z]] . M[rows[z]] for every rating z (pcrpp.cpp:24-33, :266-271).
// Rating-parallel and perfectly balanced whatever the user-length skew: a workgroup owns
// (BLOCK/G) * tile consecutive ratings, stages their (user, item) ids in LDS, and each lane group
// walks `tile` consecutive ratings 8 at a time (8 rows of M in flight per group).  Consecutive
// ratings share their user, so the u_i chunk stays in registers and is reloaded only at a user
// boundary.
// ---------------------------------------------------------------------------------------
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sddmm(const T* __restrict__ U, const T* __restrict__ M,
                                                 const int32_t* __restrict__ ruser, const int32_t* __restrict__ rows,
                                                 int64_t nnz, T* __restrict__ out, Geo geo, int tile, const int* skip,
                                                 const int32_t* __restrict__ perm, const int2* __restrict__ blk_map,
                                                 const int32_t* __restrict__ chunk_ptr) {
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    const int G = geo.G, g = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = BLOCK / G;
    const int span = ngrp * tile;
    int32_t* s_row = reinterpret_cast<int32_t*>(smem);
    int32_t* s_usr = s_row + span;
    // Default: workgroup b takes ratings [b * span, ...).  With blk_map (the XCD-aware workgroup -> chunk map of k_spmm) it
    // takes the chunks of that map instead: the same kernel then walks the tile-major CSC, where the "sequential" side
    // (U here) is the item table, the gathered side (M) the users of one L2-sized tile, and the result goes to
    // out[perm[z]] -- for item tables far beyond the L2s (Yahoo-shaped data) the gather is then served by one XCD's L2.
    int64_t b0 = (int64_t)blockIdx.x * span;
    int nb = (int)((nnz - b0 < span) ? (nnz - b0) : span);
    if (blk_map) {
        const int2 bc = blk_map[blockIdx.x];
        if (bc.y == 0) return;
        b0 = chunk_ptr[bc.x];
        nb = chunk_ptr[bc.x + bc.y] - (int)b0;
    }
    for (int t = threadIdx.x; t < nb; t += BLOCK) { s_row[t] = rows[b0 + t]; s_usr[t] = ruser[b0 + t]; }
    __syncthreads();
    const int l0 = grp * tile;
    const int l1 = (l0 + tile < nb) ? l0 + tile : nb;
    const int rho = 4 * (g & 1) + (g & 2) + ((g >> 2) & 1);
    for (int k = 0; k * G < geo.nchunk; ++k) {
        const int ch = g + k * G;
        const bool act = ch < geo.nchunk;
        int cur = -1;
        V uv = V{};
        const int chv = act ? ch : 0;                 // idle lanes re-read chunk 0 and multiply it by zero
        for (int q0 = l0; q0 < l1; q0 += PCR_UNR) {
            V rv[PCR_UNR];
            T part[PCR_UNR];
            bool fast = false;
            if (q0 + PCR_UNR <= l1) {
                // full batch: the 8 item ids and 8 user ids come in four 16-byte LDS reads, the 8 row loads are
                // issued back to back with no per-row wait or branch
                const int4 i0 = *reinterpret_cast<const int4*>(s_row + q0), i1 = *reinterpret_cast<const int4*>(s_row + q0 + 4);
                const int4 u0 = *reinterpret_cast<const int4*>(s_usr + q0), u1 = *reinterpret_cast<const int4*>(s_usr + q0 + 4);
                const int ri[PCR_UNR] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
                for (int e = 0; e < PCR_UNR; ++e) rv[e] = *reinterpret_cast<const V*>(M + row_off(ri[e], geo) + chv * VEC);
                if (u0.x != cur) {                      // at most one reload in front of a same-user batch
                    cur = u0.x;
                    uv = *reinterpret_cast<const V*>(U + row_off(cur, geo) + chv * VEC);
                    if (!act) uv = V{};
                }
                fast = (u0.y == cur) & (u0.z == cur) & (u0.w == cur) & (u1.x == cur) & (u1.y == cur) & (u1.z == cur) & (u1.w == cur);
                if (fast) {
#pragma unroll
                    for (int e = 0; e < PCR_UNR; ++e) part[e] = vdot(rv[e], uv);
                } else {                                // a user boundary inside the batch
                    const int ui8[PCR_UNR] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
                    for (int e = 0; e < PCR_UNR; ++e) {
                        if (ui8[e] != cur) {
                            cur = ui8[e];
                            uv = *reinterpret_cast<const V*>(U + row_off(cur, geo) + chv * VEC);
                            if (!act) uv = V{};
                        }
                        part[e] = vdot(rv[e], uv);
                    }
                    fast = true;
                }
            }
            if (!fast) {                                // ragged tail of the tile
#pragma unroll
                for (int e = 0; e < PCR_UNR; ++e)
                    if (q0 + e < l1) rv[e] = *reinterpret_cast<const V*>(M + row_off(s_row[q0 + e], geo) + chv * VEC);
#pragma unroll
                for (int e = 0; e < PCR_UNR; ++e) {
                    part[e] = (T)0;
                    if (q0 + e < l1) {                              // uniform inside a lane group
                        const int uu = s_usr[q0 + e];
                        if (uu != cur) {
                            cur = uu;
                            uv = *reinterpret_cast<const V*>(U + row_off(uu, geo) + chv * VEC);
                            if (!act) uv = V{};
                        }
                        part[e] = vdot(rv[e], uv);
                    }
                }
            }
            if (G >= 8) {
                const T tot = group_reduce8<T>(part, g, G);
                const int q = q0 + rho;
                if (g < 8 && q < l1) { const int64_t o = perm ? (int64_t)perm[b0 + q] : b0 + q; out[o] = (k == 0) ? tot : out[o] + tot; }
            } else {
#pragma unroll
                for (int e = 0; e < PCR_UNR; ++e) {
                    T v = part[e];
                    if (G > 2) v += __shfl_xor(v, 2);
                    if (G > 1) v += __shfl_xor(v, 1);
                    if (g == 0 && q0 + e < l1) { const int64_t o = perm ? (int64_t)perm[b0 + q0 + e] : b0 + q0 + e; out[o] = (k == 0) ? v : out[o] + v; }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_prepare: sort one user's scores (m_in, CSR order, from k_sddmm) by (level, m), write the sorted
// state, per-user loss.   One workgroup per user.
//   BIG = false: n-sized arrays in LDS;  BIG = true: in a per-workgroup global scratch slice.
// ---------------------------------------------------------------------------------------
template <typename T, bool BIG> struct LiSel { typedef uint32_t type; };
template <typename T> struct LiSel<T, true> { typedef uint64_t type; };

template <typename T>
static inline size_t prepare_bytes(int cap, int cap_pad, int rs_cap, int li_bytes) {
    // only the sort arrays need the power-of-two padding
    return carve_bytes(cap_pad, sizeof(T)) + carve_bytes(cap_pad, li_bytes) + carve_bytes(cap + 1, 8) + carve_bytes(rs_cap, 4);
}

#ifdef PCR_PREP_PROF
__device__ unsigned long long g_prep_prof[4 * 8];
#define PPROF(ph) do { if (threadIdx.x == 0) { const long long now_ = clock64(); pp_[ph] += now_ - pt_; pt_ = now_; } } while (0)
#else
#define PPROF(ph) do { } while (0)
#endif
// body of k_prepare for a team of BLOCK threads (smem: the team's LDS): team `first` of `step` walks users first, first + step, ...
template <typename T, int BLOCK, bool BIG>
__device__ __forceinline__ void prepare_body(char* smem, const Shard<T>& S, const int32_t* __restrict__ users, int nusers,
                                             const T* __restrict__ m_in, int cap, int cap_pad, int rs_cap, char* scratch,
                                             size_t stride, int strict, int first, int step) {
    typedef typename LiSel<T, BIG>::type LI;
    Carver small(smem);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    Carver big(BIG ? scratch + (size_t)first * stride : small.p);
    T* key = big.take<T>(cap_pad);
    LI* li = big.take<LI>(cap_pad);
    double* Sx = big.take<double>(cap + 1);
    int* rs = big.take<int>(rs_cap);
    const int tid = btid<BLOCK>();
#ifdef PCR_PREP_PROF
    long long pp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_ = clock64();
#endif

    for (int ui = first; ui < nusers; ui += step) {
        const int u = users[ui];
        const int64_t s0 = S.uptr[u];
        const int n = (int)(S.uptr[u + 1] - s0);
        const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
        if (n == 0) {
            if (tid == 0) S.objp[u] = 0.0;
            continue;
        }
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = S.runstart[S.runofs[u] + l];
        const int npad = next_pow2(n);
        // start from the user's PREVIOUS order when there is one (any earlier sorted state: levels are static, so it is level-
        // grouped already, and from the third iteration on nearly (level, m)-sorted for the new scores too: resort_window)
        const bool from_prev = !BIG && S.resort_d > 0 && S.prev_valid;
#pragma unroll 4
        for (int p = tid; p < npad; p += BLOCK) {
            if (p < n) {
                if (from_prev) { const unsigned idx = (unsigned)S.sidx[s0 + p]; li[p] = LiOps<LI>::pack(S.slvl[s0 + p], idx); key[p] = m_in[s0 + idx]; }
                else { li[p] = LiOps<LI>::pack(S.lvl[s0 + p], (unsigned)p); key[p] = m_in[s0 + p]; }
            } else { li[p] = LiOps<LI>::pack(0xFFFFu, (unsigned)p); key[p] = (T)0; }
        }
        bsync<BLOCK>();
        PPROF(0);
        // (tmp and the second key array share the prefix-sum array, idle until the loss: 4-byte scores only -- fp64 sorts fully)
        bool resorted = false;
        if constexpr (!BIG)
            if (from_prev)
                resorted = resort_window<T, LI, BLOCK>(key, li, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, n, reinterpret_cast<int*>(Sx),
                                                       sizeof(T) == 4 ? reinterpret_cast<T*>(reinterpret_cast<int*>(Sx) + n) : (T*)nullptr, S.resort_d, reinterpret_cast<int*>(red));
        if (!resorted) bitonic_sort<T, LI, BLOCK, false, !BIG>(key, li, npad);
        PPROF(1);
        for (int p = tid; p < n; p += BLOCK) {
            const LI x = li[p];
            const unsigned idx = LiOps<LI>::idx(x);
            S.ms[s0 + p] = key[p];
            S.slvl[s0 + p] = (uint16_t)LiOps<LI>::lev(x);
            S.sitem[s0 + p] = S.item[s0 + idx];
            S.sidx[s0 + p] = (int32_t)idx;
        }
        PPROF(2);
        double loss;
        if (S.ws) {
            for (int p = tid; p < n; p += BLOCK)
                store_windows<T>(S, (size_t)s0 + p, key, rs, nlev, (int)LiOps<LI>::lev(li[p]), key[p], strict);
            PPROF(3);
            auto levf = [&](int p) { return (int)LiOps<LI>::lev(li[p]); };
            loss = S.w16 ? block_objective_win<T, BLOCK>(key, levf, rs, nlev, n, reinterpret_cast<const uint16_t*>(S.win) + (size_t)s0 * S.ws, S.ws, Sx, red)
                         : block_objective_win<T, BLOCK>(key, levf, rs, nlev, n, reinterpret_cast<const uint32_t*>(S.win) + (size_t)s0 * S.ws, S.ws, Sx, red);
        } else {
            loss = block_objective<T, BLOCK>(key, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, nlev, n, Sx, red, strict);
        }
        if (tid == 0) S.objp[u] = loss;
        bsync<BLOCK>();
        PPROF(4);
    }
#ifdef PCR_PREP_PROF
    if (tid == 0) {
        const int cls = BLOCK == 64 ? 0 : BLOCK == 256 ? 1 : BIG ? 3 : 2;
        for (int i = 0; i < 5; ++i) atomicAdd(&g_prep_prof[cls * 8 + i], (unsigned long long)pp_[i]);
        atomicAdd(&g_prep_prof[cls * 8 + 7], 1ull);
    }
#endif
}
template <typename T, int BLOCK, bool BIG>
__global__ __launch_bounds__(BLOCK) void k_prepare(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                   const T* __restrict__ m_in,
                                                   int cap, int cap_pad, int rs_cap, char* scratch, size_t stride, int strict) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    prepare_body<T, BLOCK, BIG>(smem, S, users, nusers, m_in, cap, cap_pad, rs_cap, scratch, stride, strict, (int)blockIdx.x, (int)gridDim.x);
}

// Both LDS-resident classes in ONE launch of 512-thread workgroups: workgroups [0, nblk_b) take the long users of list B
// one per workgroup, the others eight short users of list A each, one per wave (a one-wave team needs no workgroup
// barrier).  One launch instead of one per class on concurrent streams: no fork / join around the line search.
template <typename T, int WB>
__global__ __launch_bounds__(WB) void k_prepare_all(Shard<T> S, const int32_t* __restrict__ users_a, int nusers_a, int cap_a,
                                                     int cap_pad_a, int rs_cap_a, size_t wave_bytes,
                                                     const int32_t* __restrict__ users_b, int nusers_b, int cap_b, int cap_pad_b,
                                                     int rs_cap_b, int nblk_b, const T* __restrict__ m_in, int strict) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nblk_b)
        prepare_body<T, WB, false>(smem, S, users_b, nusers_b, m_in, cap_b, cap_pad_b, rs_cap_b, nullptr, 0, strict,
                                   (int)blockIdx.x, nblk_b);
    else
        prepare_body<T, 64, false>(smem + (size_t)(threadIdx.x >> 6) * wave_bytes, S, users_a, nusers_a, m_in, cap_a, cap_pad_a,
                                   rs_cap_a, nullptr, 0, strict, ((int)blockIdx.x - nblk_b) * (WB / 64) + (int)(threadIdx.x >> 6),
                                   ((int)gridDim.x - nblk_b) * (WB / 64));
}

// ---------------------------------------------------------------------------------------
// k_vsweep: per-user sweep coefficients for the V side (written in sorted order).
//   HV = false: gradient (x = m, shift 1)           pcrpp.cpp:214-238
//   HV = true : Hessian-vector (x = b = u_i . a_item, computed by k_sddmm)   pcrpp.cpp:294-318
// ---------------------------------------------------------------------------------------
template <typename T>
static inline size_t vsweep_bytes(int cap, int rs_cap, bool two) {      // two: scores AND sweep values (HV without window cache)
    return carve_bytes(cap, sizeof(T)) * (two ? 2 : 1) + carve_bytes(cap + 1, 8) + carve_bytes(rs_cap, 4);
}

// body of k_vsweep: workgroup `blk` of `nblk` walks users blk, blk + nblk, ...
template <typename T, int BLOCK, bool BIG, bool HV>
__device__ __forceinline__ void vsweep_block_body(char* smem, const Shard<T>& S, const int32_t* __restrict__ users, int nusers,
                                                  const T* __restrict__ bsrc, T* __restrict__ c_out, int cap, int rs_cap,
                                                  char* scratch, size_t stride, int strict, int blk, int nblk, int flags = 0) {
    const int b_csr = flags & 1, pf4 = flags & 2;       // b in CSR order; four rounds of per-rating loads in flight
    Carver small(smem);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    Carver big(BIG ? scratch + (size_t)blk * stride : small.p);
    T* ms = big.take<T>(cap);                           // scores (thresholds) -- not loaded when the window cache replaces them
    T* x = (HV && !S.ws) ? big.take<T>(cap) : ms;       // sweep values b; they share the array unless both are needed
    double* Sx = big.take<double>(cap + 1);
    int* rs = big.take<int>(rs_cap);
    const int tid = threadIdx.x;

    for (int ui = blk; ui < nusers; ui += nblk) {
        const int u = users[ui];
        const int64_t s0 = S.uptr[u];
        const int n = (int)(S.uptr[u + 1] - s0);
        const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
        if (n == 0) continue;
        if (!HV || !S.ws) {
#pragma unroll 4
            for (int p = tid; p < n; p += BLOCK) ms[p] = S.ms[s0 + p];
        }
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = S.runstart[S.runofs[u] + l];
        const T* xs = ms;
        if (HV) {     // b = u_i . a_item from k_sddmm: in sorted order when it walked sitem, in CSR order when it walked the CSC
#pragma unroll 4
            for (int p = tid; p < n; p += BLOCK) x[p] = bsrc[s0 + (b_csr ? S.sidx[s0 + p] : p)];
            xs = x;
        }
        __syncthreads();
        block_excl_scan<BLOCK>([&](int i) { return (double)xs[i]; }, Sx, n, red);
        if (S.ws == 4 && S.w16 && pf4) { sweep_out4<T, BLOCK, HV>(S, s0, n, nlev, tid, xs, Sx, rs, c_out); __syncthreads(); continue; }
        for (int p = tid; p < n; p += BLOCK) {
            const int lev = S.slvl[s0 + p];
            const double c = S.ws
                ? sweep_coeff_cached<T>(S, (size_t)(s0 + p), Sx, rs, nlev, lev, (double)xs[p], HV ? 0.0 : 1.0)
                : sweep_coeff<T>(ms, Sx, rs, nlev, lev, ms[p], (double)xs[p], HV ? 0.0 : 1.0, strict);
            // c goes out in CSR order -- a permutation inside this user's own segment, so the lines it touches are written in
            // full by this workgroup (staging the permutation through LDS was measured slower: 2.11 against 1.65 ms per
            // launch on the Netflix shape, the extra array costs occupancy)
            c_out[s0 + S.sidx[s0 + p]] = (T)c;
        }
        __syncthreads();
    }
}
template <typename T, int BLOCK, bool BIG, bool HV>
__global__ __launch_bounds__(BLOCK) void k_vsweep(Shard<T> S, Geo geo, const int32_t* __restrict__ users, int nusers,
                                                  const T* __restrict__ bsrc, T* __restrict__ c_out,
                                                  int cap, int rs_cap, char* scratch, size_t stride, int strict, const int* skip,
                                                  int b_csr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    vsweep_block_body<T, BLOCK, BIG, HV>(smem, S, users, nusers, bsrc, c_out, cap, rs_cap, scratch, stride, strict,
                                         (int)blockIdx.x, (int)gridDim.x, b_csr);
}

// k_vsweep for short users (<= 256 ratings), ONE WAVE PER USER, four users per 256-thread
// workgroup: no workgroup barriers at all (LDS traffic stays inside a wave, which the LDS serves
// in program order), so the many short users of a rating set do not pay a block's fixed cost each.
template <typename T>
static inline size_t vsweep_wave_bytes(int cap, int rs_cap, bool two) {      // per wave
    return carve_bytes(cap, sizeof(T)) * (two ? 2 : 1) + carve_bytes(cap + 1, 8) + carve_bytes(rs_cap, 4);
}
// body: this wave sweeps user number ui of the list
template <typename T, bool HV>
__device__ __forceinline__ void vsweep_wave_body(char* smem, const Shard<T>& S, const int32_t* __restrict__ users, int nusers,
                                                 const T* __restrict__ bsrc, T* __restrict__ c_out,
                                                 int cap, int rs_cap, size_t wave_bytes, int strict, int ui, int flags = 0) {
    const int b_csr = flags & 1, pf4 = flags & 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (ui >= nusers) return;
    Carver big(smem + (size_t)wid * wave_bytes);
    T* ms = big.take<T>(cap);                           // scores (thresholds) -- not loaded when the window cache replaces them
    T* x = (HV && !S.ws) ? big.take<T>(cap) : ms;       // sweep values b; they share the array unless both are needed
    double* Sx = big.take<double>(cap + 1);
    int* rs = big.take<int>(rs_cap);
    const int u = users[ui];
    const int64_t s0 = S.uptr[u];
    const int n = (int)(S.uptr[u + 1] - s0);
    const int nlev = (int)(S.runofs[u + 1] - S.runofs[u]) - 1;
    if (n == 0) return;
    if (!HV || !S.ws) {
#pragma unroll 4
        for (int p = lane; p < n; p += 64) ms[p] = S.ms[s0 + p];
    }
    for (int l = lane; l <= nlev; l += 64) rs[l] = S.runstart[S.runofs[u] + l];
    const T* xs = ms;
    if (HV) {
#pragma unroll 4
        for (int p = lane; p < n; p += 64) x[p] = bsrc[s0 + (b_csr ? S.sidx[s0 + p] : p)];
        xs = x;
    }
    wave_sync();
    double carry = 0.0;                                     // wave-level exclusive scan of xs -> Sx[0..n]
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const double v = (i < n) ? (double)xs[i] : 0.0;
        const double inc = wave_incl_scan(v);
        if (i < n) Sx[i] = carry + inc - v;
        carry += lane63(inc);
    }
    if (lane == 0) Sx[n] = carry;
    wave_sync();
    if (S.ws == 4 && S.w16 && pf4) { sweep_out4<T, 64, HV>(S, s0, n, nlev, lane, xs, Sx, rs, c_out); return; }
    for (int p = lane; p < n; p += 64) {
        const int lev = S.slvl[s0 + p];
        const double c = S.ws
            ? sweep_coeff_cached<T>(S, (size_t)(s0 + p), Sx, rs, nlev, lev, (double)xs[p], HV ? 0.0 : 1.0)
            : sweep_coeff<T>(ms, Sx, rs, nlev, lev, ms[p], (double)xs[p], HV ? 0.0 : 1.0, strict);
        c_out[s0 + S.sidx[s0 + p]] = (T)c;
    }
}
template <typename T, bool HV>
__global__ __launch_bounds__(256) void k_vsweep_wave(Shard<T> S, const int32_t* __restrict__ users, int nusers,
                                                     const T* __restrict__ bsrc, T* __restrict__ c_out,
                                                     int cap, int rs_cap, size_t wave_bytes, int strict, const int* skip, int b_csr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    vsweep_wave_body<T, HV>(smem, S, users, nusers, bsrc, c_out, cap, rs_cap, wave_bytes, strict,
                            (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), b_csr);
}

// Both LDS-resident classes in ONE launch of 512-thread workgroups (the two sweeps are each shorter than a launch
// round trip, so back to back they cost two kernel latencies and side by side a fork/join): workgroups [0, nblk_b)
// take the long users of list B one per workgroup, the others take eight short users of list A, one per wave.
template <typename T, bool HV, int WB>
__global__ __launch_bounds__(WB) void k_vsweep_all(Shard<T> S, const int32_t* __restrict__ users_a, int nusers_a, int cap_a,
                                                    int rs_cap_a, size_t wave_bytes, const int32_t* __restrict__ users_b,
                                                    int nusers_b, int cap_b, int rs_cap_b, int nblk_b,
                                                    const T* __restrict__ bsrc, T* __restrict__ c_out, int strict, const int* skip, int b_csr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if ((int)blockIdx.x < nblk_b)
        vsweep_block_body<T, WB, false, HV>(smem, S, users_b, nusers_b, bsrc, c_out, cap_b, rs_cap_b, nullptr, 0, strict,
                                            (int)blockIdx.x, nblk_b, b_csr);
    else
        vsweep_wave_body<T, HV>(smem, S, users_a, nusers_a, bsrc, c_out, cap_a, rs_cap_a, wave_bytes, strict,
                                ((int)blockIdx.x - nblk_b) * (WB / 64) + (int)(threadIdx.x >> 6), b_csr);
}

// ---------------------------------------------------------------------------------------
// k_spmm + k_spmm_fin: out[j,:] = beta * base[j,:] + sum_{z in column j} c[z] * U[cuser[z],:]
// (pcrpp.cpp:240-243, 323-327).  Item-major (CSC) gather instead of the reference's per-scalar
// atomics, and NO atomics at all: the CSC nnz range is cut into equal chunks (load balance
// independent of item popularity); a group of G lanes walks one chunk, keeps the running row in
// fp64 registers and stores ONE partial row per (chunk, item) incidence with plain coalesced
// stores into a slab whose slot numbering is static (slots of one item are consecutive).
// k_spmm_fin then sums each item's slots in a fixed order -> bitwise reproducible.
// c (CSR order) is read through c2r, the STATIC CSC entry -> CSR position map of the shard.
// XCD-aware tiling: the CSC is built per USER TILE (a contiguous user range whose rows of U, and whose slice of c, fit
// one XCD's 4 MB L2), tile-major, and workgroup b works on a tile t with t % 8 == b % 8 (workgroups go to the XCDs
// round-robin), so the random row gather of a tile is served by ONE L2 instead of every L2 holding a copy of all of U
// and c.  Chunks never straddle tiles (chunk_ptr); an item's slab slots are consecutive whatever tile they come from
// (slot_id maps the (chunk, item) incidences, enumerated in chunk order, to item-major slab rows).
// ---------------------------------------------------------------------------------------
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_spmm(const T* __restrict__ c, const int32_t* __restrict__ c2r,
                                                const int32_t* __restrict__ cuf,
                                                const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ inc_base,
                                                const int32_t* __restrict__ slot_id, const int2* __restrict__ blk_chunks,
                                                const T* __restrict__ U, T* __restrict__ slab, Geo geo, const int* skip) {
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    if (skip && *skip) return;
    const int G = geo.G, g = threadIdx.x & (G - 1);
    const int2 bc = blk_chunks[blockIdx.x];                    // first chunk and number of chunks of this workgroup
    if ((int)threadIdx.x / G >= bc.y) return;
    const int gid = bc.x + (int)threadIdx.x / G;
    const int64_t z0 = chunk_ptr[gid], z1 = chunk_ptr[gid + 1];
    for (int k = 0; k * G < geo.nchunk; ++k) {
        const int ch = g + k * G;
        const bool act = ch < geo.nchunk;
        // The running row of a chunk (<= 128 terms) is kept in T: it is rounded to T when it is stored into the slab anyway,
        // and for T = float the fp64 multiply-adds were a fifth of this kernel's time (39 -> 35 us); k_spmm_fin adds the
        // slab rows of an item in fp64.
        T acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = (T)0;
        int inc = inc_base[gid];
        auto flush = [&]() {
            if (act) {
                V o;
                T* op = reinterpret_cast<T*>(&o);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { op[e] = (T)acc[e]; acc[e] = (T)0; }
                {   // non-temporal: the slab is written once and read once by k_spmm_fin; streaming stores leave fewer dirty lines
                    // for the L2 write-back at the kernel boundary (40 ml1m iterations 71.0-71.2 -> 70.0-70.6 ms)
                    typedef T nat __attribute__((ext_vector_type(VEC)));
                    nat ov;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ov[e] = op[e];
                    __builtin_nontemporal_store(ov, reinterpret_cast<nat*>(slab + row_off(slot_id[inc], geo) + ch * VEC));
                }
            }
            inc += 1;
        };
        for (int64_t zb = z0; zb < z1; zb += G) {
            const int64_t zi = zb + g;
            T cr = (T)0;
            int ur = 0;
            if (zi < z1) { cr = c2r ? c[c2r[zi]] : c[zi]; ur = cuf[zi]; }            // user id; sign bit: a new item starts here (not at the chunk start)
            const int cnt = (int)((z1 - zb < G) ? (z1 - zb) : G);
            for (int q = 0; q < cnt; q += PCR_UNR) {
                V rv[PCR_UNR];
                T cc[PCR_UNR];
                int uf[PCR_UNR];
#pragma unroll
                for (int e8 = 0; e8 < PCR_UNR; ++e8) {
                    if (q + e8 < cnt) {
                        cc[e8] = __shfl(cr, q + e8, G);
                        uf[e8] = __shfl(ur, q + e8, G);
                        if (act) rv[e8] = *reinterpret_cast<const V*>(U + row_off(uf[e8] & 0x7fffffff, geo) + ch * VEC);
                    }
                }
#pragma unroll
                for (int e8 = 0; e8 < PCR_UNR; ++e8) {
                    if (q + e8 < cnt) {
                        if (uf[e8] < 0) flush();
                        if (act) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[e] += cc[e8] * velem(rv[e8], e);
                        }
                    }
                }
            }
        }
        flush();
    }
}

// out[j,:] = beta * base[j,:] + sum of the item's slab slots [item_slot[j], item_slot[j+1]); G lanes per item, items
// strided over the grid.  DOTS (CG on one GPU, where out = Hp is final here and base = p): the kernel also leaves the
// partials of p.Hp, rr.p, rr.Hp and Hp.Hp in part[blk][4] -- everything the CG scalars of this iteration need
// (k_cg_bc) -- so no separate pass re-reads p, Hp, rr.
template <typename T, int BLOCK, bool DOTS>
__global__ __launch_bounds__(BLOCK) void k_spmm_fin(const T* __restrict__ slab, const int32_t* __restrict__ item_slot,
                                                    const T* __restrict__ base, double beta, int d2, T* __restrict__ out,
                                                    Geo geo, const int* skip, const T* __restrict__ rr, double* __restrict__ part,
                                                    int j0 = 0) {                      // items [j0, d2)
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    __shared__ double red[BLOCK / PCR_WAVE + 1];
    if (skip && *skip) return;
    const int G = geo.G, g = threadIdx.x & (G - 1), ipb = BLOCK / G;
    double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
    for (int j = j0 + (int)blockIdx.x * ipb + (int)threadIdx.x / G; j < d2; j += (int)gridDim.x * ipb) {
        const int s0 = item_slot[j], s1 = item_slot[j + 1];
        for (int ch = g; ch < geo.nchunk; ch += G) {
            double acc[VEC];
            const V bv = *reinterpret_cast<const V*>(base + row_off(j, geo) + ch * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = beta * (double)velem(bv, e);
            // the item's slab rows are consecutive: eight loads in flight, added in slot order (the same sum as one by one)
            for (int sl = s0; sl < s1; sl += 8) {
                V pv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (sl + q < s1) pv[q] = *reinterpret_cast<const V*>(slab + row_off(sl + q, geo) + ch * VEC);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (sl + q < s1) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[e] += (double)velem(pv[q], e);
                    }
            }
            V o;
            T* op = reinterpret_cast<T*>(&o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) op[e] = (T)acc[e];
            *reinterpret_cast<V*>(out + row_off(j, geo) + ch * VEC) = o;
            if (DOTS) {
                const V rv = *reinterpret_cast<const V*>(rr + row_off(j, geo) + ch * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const double pe = (double)velem(bv, e), he = (double)op[e], re = (double)velem(rv, e);
                    x += pe * he;
                    y += re * pe;
                    z += re * he;
                    w += he * he;
                }
            }
        }
    }
    if (DOTS) {
        x = block_sum<BLOCK>(x, red);
        y = block_sum<BLOCK>(y, red);
        z = block_sum<BLOCK>(z, red);
        w = block_sum<BLOCK>(w, red);
        if (threadIdx.x == 0) { part[4 * blockIdx.x] = x; part[4 * blockIdx.x + 1] = y; part[4 * blockIdx.x + 2] = z; part[4 * blockIdx.x + 3] = w; }
    }
}

// stream calibration (pcr_solver.hip, pick_lanes): hold a hardware queue busy for `ticks` of the constant-rate clock
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 22) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}
__global__ void k_nop() {}

// ---------------------------------------------------------------------------------------
// elementwise / CG kernels (solve_delta_new, pcrpp.cpp:335-358).  Scalars stay on the device;
// every reduction is two-stage and deterministic (per-block partials, then each consumer
// block re-reduces the short partial array in a fixed order).
// ---------------------------------------------------------------------------------------
struct CGState {
    double g2, err, pHp, rp, alpha, rr2, rHp, beta;
    double rr2buf[2];      // |rr|^2 after iteration k lives in rr2buf[k & 1] (double-buffered: readers and the writer of one launch never share a slot)
    int done, iters;
    int done_at, pad_;     // iteration whose update met the stop test (0: none yet)
};

#define PCR_EW_BLOCK 256

template <typename T>
__global__ void k_axpy_out(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b, double s, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // out = a + s*b (mat_substract_vec, util.cpp:395)
    if (i < n) out[i] = (T)((double)a[i] + s * (double)b[i]);
}

__device__ __forceinline__ void reduce_partials2(const double* part, int nblk, double* a, double* b, double* red) {
    // every block reduces the (short) partial array identically: deterministic
    double x = 0.0, y = 0.0;
    for (int i = threadIdx.x; i < nblk; i += PCR_EW_BLOCK) { x += part[2 * i]; y += part[2 * i + 1]; }
    x = block_sum<PCR_EW_BLOCK>(x, red);
    y = block_sum<PCR_EW_BLOCK>(y, red);
    *a = x; *b = y;
}

// sum of squares of a (and optionally dot(a, b)) -> part[blk][2]
template <typename T>
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_dots(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                        int per_block, double* __restrict__ part) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double x = 0.0, y = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) {
        const double av = (double)a[i];
        x += av * av;
        if (b) y += av * (double)b[i];
    }
    x = block_sum<PCR_EW_BLOCK>(x, red);
    y = block_sum<PCR_EW_BLOCK>(y, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = x; part[2 * blockIdx.x + 1] = y; }
}

// out[0] = sum part[2i], out[1] = sum part[2i+1]
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_fin2(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    double a, b;
    reduce_partials2(part, nblk, &a, &b, red);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

// plain sum of a double array, two-stage
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_sum_stage1(const double* __restrict__ in, int64_t n, int per_block,
                                                              double* __restrict__ part) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double x = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) x += in[i];
    x = block_sum<PCR_EW_BLOCK>(x, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = x; part[2 * blockIdx.x + 1] = 0.0; }
}

// CG start: delta = 0, rr = -g, p = g; partial |g|^2
template <typename T>
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_cg_init(const T* __restrict__ g, T* __restrict__ delta, T* __restrict__ rr,
                                                           T* __restrict__ p, int64_t n,
                                                           int per_block, double* __restrict__ part) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double x = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) {
        const T gv = g[i];
        delta[i] = (T)0;
        rr[i] = -gv;
        p[i] = gv;
        x += (double)gv * (double)gv;
    }
    x = block_sum<PCR_EW_BLOCK>(x, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = x; part[2 * blockIdx.x + 1] = 0.0; }
}

__global__ __launch_bounds__(PCR_EW_BLOCK) void k_cg_init_fin(const double* __restrict__ part, int nblk, CGState* st, double tol) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    double a, b;
    reduce_partials2(part, nblk, &a, &b, red);
    if (threadIdx.x == 0) {
        st->g2 = a;
        st->err = sqrt(a) * tol;             // pcrpp.cpp:340 (tol = 0.01 there)
        st->done = 0;
        st->iters = 0;
        st->done_at = 0;
        st->rr2buf[0] = a;                   // rr = -g
    }
}

// A (only when an all-reduce sits between k_spmm_fin and the dot products): partials of p.Hp, rr.p, rr.Hp, Hp.Hp
template <typename T>
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_cg_a(const T* __restrict__ p, const T* __restrict__ Hp, const T* __restrict__ rr,
                                                        int64_t n, int per_block, double* __restrict__ part,
                                                        CGState* st) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    if (st->done) return;     // CG already converged: later iterations are queued but idle
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) {
        const double pv = (double)p[i], h = (double)Hp[i], r = (double)rr[i];
        x += pv * h; y += r * pv; z += r * h; w += h * h;
    }
    x = block_sum<PCR_EW_BLOCK>(x, red);
    y = block_sum<PCR_EW_BLOCK>(y, red);
    z = block_sum<PCR_EW_BLOCK>(z, red);
    w = block_sum<PCR_EW_BLOCK>(w, red);
    if (threadIdx.x == 0) { part[4 * blockIdx.x] = x; part[4 * blockIdx.x + 1] = y; part[4 * blockIdx.x + 2] = z; part[4 * blockIdx.x + 3] = w; }
}

// B + C of iteration k in ONE pass (pcrpp.cpp:346-356): alpha = -(rr.p)/(p.Hp); delta += alpha p; rr += alpha Hp; stop if
// |rr| < err, else beta = (rr.Hp)/(p.Hp), p = -rr + beta p.  The two dot products of the NEW residual follow from the four
// of the old one without touching the vectors again,
//     |rr + a Hp|^2 = |rr|^2 + 2 a rr.Hp + a^2 Hp.Hp,      (rr + a Hp).Hp = rr.Hp + a Hp.Hp,
// so beta and the stop test are known before the update and one kernel does what took two (and a grid-wide reduction
// between them).  Every block reduces the short partial array identically: deterministic.
// EXACT (a caller-set cg_tol below 1e-5): the recurrence for |rr|^2 cancels catastrophically once the residual has dropped
// by many orders of magnitude, so the stop test is taken on the directly summed |rr_new|^2 instead: this kernel leaves its
// partials in part_rr and always updates p, k_cg_stop (one block) decides.
template <typename T, bool EXACT>
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_cg_bc(T* __restrict__ p, const T* __restrict__ Hp, T* __restrict__ rr,
                                                         T* __restrict__ delta, int64_t n, int per_block, int nblk,
                                                         const double* __restrict__ part, CGState* st, int k,
                                                         double* __restrict__ part_rr) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    // CG already converged in an EARLIER launch: later iterations are queued but idle.  (done_at == k can only have been
    // written by the last block of THIS launch: a block that starts late must still do its slice.)
    const int da = st->done_at;
    if (da != 0 && da < k) return;
    double s4[4];
    for (int c = 0; c < 4; ++c) {
        double x = 0.0;
        for (int i = threadIdx.x; i < nblk; i += PCR_EW_BLOCK) x += part[4 * i + c];
        s4[c] = block_sum<PCR_EW_BLOCK>(x, red);
    }
    const double pHp = s4[0], rp = s4[1], rHp0 = s4[2], HpHp = s4[3];
    const double rr2_old = st->rr2buf[(k - 1) & 1];
    const double alpha = -1.0 * rp / pHp;
    double rr2 = rr2_old + 2.0 * alpha * rHp0 + alpha * alpha * HpHp;
    rr2 = rr2 > 0.0 ? rr2 : 0.0;
    const double rHp = rHp0 + alpha * HpHp;
    const bool conv = !EXACT && sqrt(rr2) < st->err;        // pcrpp.cpp:350
    const double beta = rHp / pHp;
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double x2 = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) {
        const double pv = (double)p[i], hv = (double)Hp[i];
        delta[i] = (T)((double)delta[i] + pv * alpha);
        const T rn = (T)((double)rr[i] + hv * alpha);
        rr[i] = rn;
        if (EXACT) x2 += (double)rn * (double)rn;
        if (!conv) p[i] = (T)((double)rn * -1.0 + pv * beta);
    }
    if (EXACT) {
        x2 = block_sum<PCR_EW_BLOCK>(x2, red);
        if (threadIdx.x == 0) { part_rr[2 * blockIdx.x] = x2; part_rr[2 * blockIdx.x + 1] = 0.0; }
    }
    // The host queues all 10 iterations without waiting; once `done` is set every later kernel of the solve returns at
    // once.  Blocks of THIS launch read done_at (see above) and rr2buf[(k-1)&1] only.
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        st->pHp = pHp; st->rp = rp; st->alpha = alpha; st->rr2 = rr2; st->rHp = rHp; st->beta = beta;
        st->iters += 1;
        if (!EXACT) {
            st->rr2buf[k & 1] = rr2;
            if (conv) { st->done_at = k; st->done = 1; }
        }
    }
}
// stop test of iteration k on the directly summed residual (k_cg_bc<EXACT>)
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_cg_stop(const double* __restrict__ part_rr, int nblk, CGState* st, int k) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    if (st->done) return;
    double a, b;
    reduce_partials2(part_rr, nblk, &a, &b, red);
    if (threadIdx.x == 0) {
        st->rr2 = a; st->rr2buf[k & 1] = a;
        if (sqrt(a) < st->err) { st->done_at = k; st->done = 1; }
    }
}

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
// workgroup barrier -> loads of the handed-off bytes (agent-scope, sc1).  That is the formally ordered form and the default.
// fenced = false (pcr_tune "cluster_fence" = 0) drops the release and the acquire: the payload is sc1 both ways, which is the
// first row of MI355X_MICROARCH.md's table of hand-offs measured valid WITHOUT the acquire on gfx950 -- measured, "not an
// architectural guarantee", and its "one workgroup per CU" cell does not hold while other length classes share the CUs -- for
// 5 % of the cluster class (ml1m: 409 -> 388 us, 1 % of a step).  Placement-independent either way; the launch keeps the grid
// <= one workgroup per CU so all members are co-resident.
// ---------------------------------------------------------------------------------------
struct ClusterBufs {
    unsigned* bar;          // one arrival counter per cluster (zeroed before every launch)
    char* xch;              // per cluster: 2 x cap_pad scores (T) + 2 x K x ld doubles
    size_t xch_stride;
    unsigned long long* rows;   // this length class's cumulative count of gathered rows (pcr_tune "count_rows"; never reset)
};

template <int K>
__device__ __forceinline__ void cluster_barrier(unsigned* bar, unsigned& phase, unsigned long long* err, bool fenced) {
    if (K == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its write-through stores
    __syncthreads();
    phase += 1;
    if (threadIdx.x == 0) {
        if (fenced) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (explicit: the compiler may drop the wait behind the write-back)
        }
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
        if (fenced) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
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
                                                 int wcap) {
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
        cluster_barrier<K>(bar, phase, counters + 3, !(fault & 16));
        for (int p = tid; p < n; p += BLOCK) key[p] = __hip_atomic_load(buf + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    };
    // vec += sum over members (fixed order) of their partial r-vectors
    auto exchange_vector = [&](double* vec) {
        if (K == 1) return;
        double* buf = (xv_par & 1) ? xv1 : xv0; xv_par += 1;
        for (int t = tid; t < ld; t += BLOCK)
            __hip_atomic_store(buf + (size_t)mem * ld + t, part[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cluster_barrier<K>(bar, phase, counters + 3, !(fault & 16));
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
            // ---- CG, solve_delta_u_new (pcrpp.cpp:628-647)
            for (int t = tid; t < ld; t += BLOCK) { delta[t] = 0.0; rr[t] = gvec[t] * -1.0; pv[t] = gvec[t]; }
            const double err = sqrt(gn2) * cg_tol;                              // 0.01 in the reference (:632)
            // The first line-search try needs no pass over the rows: V_I (u - s delta) = m - s sum_k alpha_k (V_I p_k), and
            // b_k = V_I p_k is what every CG iteration computes anyway.  With the window cache on, the gradient point's scores
            // ms0 are not needed again after the gradient sweep, so they carry the running m - s0 sum alpha_k b_k; the sweep
            // writes its coefficients beside b (into the sort's index array, idle until the line search) so that b survives
            // until alpha is known.  Not when the sorted state belongs to a REJECTED V_new (its m is not V_I u, quirk q5), not
            // without the window cache (the sweeps then search ms0), not when T is wider than the index array (fp64 in LDS).
            const bool mrec = (fault & 8) && !(fault & 4) && win && sizeof(T) <= sizeof(LI);
            T* cst = mrec ? reinterpret_cast<T*>(li) : key;
            ls_free = mrec ? 1 : 0;
            __syncthreads();
            for (int k = 1; k <= cg_max; ++k) {                                 // 10 in the reference (:636)
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
                if (!resorted) bitonic_sort<T, LI, BLOCK, false, !BIG>(key, li, npad);
                UPROF(8);
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
                if (S.ws) store_windows<T>(S, (size_t)s0 + p, key, rs, nlev, lev, key[p], strict);
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

// ---------------------------------------------------------------------------------------
// k_eval: compute_pairwise_error_ndcg (util.cpp:434-542), one workgroup per user.
//   pairwise error: #{ordered (a,b): s_a >= s_b && v_a < v_b} / (n(n-1)/2)   (util.cpp:467-483;
//     the reference's two tests on unordered pairs are this one test on ordered pairs; score
//     ties count as errors, all pairs are in the denominator)
//   NDCG@min(k,n): top-k by score (util.cpp:494-495; ties: lower index first -- the reference's
//     std::sort leaves tie order unspecified), gains 2^v - 1 and the ideal DCG are static per
//     data set and precomputed on the host with the reference's own pow()/log2() arithmetic.
// out4[u] = {err ratio, has pairs, ndcg, has ratings}
// ---------------------------------------------------------------------------------------
template <typename T>
static inline size_t eval_bytes(int cap) { return carve_bytes(cap, sizeof(T)) + carve_bytes(cap, 8) + carve_bytes(cap, 4); }

template <typename T, int BLOCK, bool BIG>
__global__ __launch_bounds__(BLOCK) void k_eval(const int64_t* __restrict__ uptr, const int32_t* __restrict__ item,
                                                const double* __restrict__ val, const double* __restrict__ gain,
                                                const double* __restrict__ idcg, const double* __restrict__ disc, int ndcg_k,
                                                const int32_t* __restrict__ users, int nusers, const T* __restrict__ U,
                                                const T* __restrict__ Vm, Geo geo, double* __restrict__ out4, int cap,
                                                char* scratch, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver small(smem);
    T* vecT = small.take<T>(geo.ld);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    T* wmax = small.take<T>(BLOCK / PCR_WAVE + 1);
    int* widx = small.take<int>(BLOCK / PCR_WAVE + 1);
    Carver big(BIG ? scratch + (size_t)blockIdx.x * stride : small.p);
    T* sc = big.take<T>(cap);
    double* vv = big.take<double>(cap);
    int32_t* itm = big.take<int32_t>(cap);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    for (int ui = blockIdx.x; ui < nusers; ui += gridDim.x) {
        const int u = users[ui];
        const int64_t s0 = uptr[u];
        const int n = (int)(uptr[u + 1] - s0);
        if (n == 0) {
            if (tid == 0) { out4[4 * (size_t)u] = 0.0; out4[4 * (size_t)u + 1] = 0.0; out4[4 * (size_t)u + 2] = 0.0; out4[4 * (size_t)u + 3] = 0.0; }
            continue;
        }
        for (int t = tid; t < geo.ld; t += BLOCK) vecT[t] = U[(size_t)u * geo.ld + t];
        for (int p = tid; p < n; p += BLOCK) { vv[p] = val[s0 + p]; itm[p] = item[s0 + p]; }
        __syncthreads();
        block_sddmm<T, BLOCK>(Vm, vecT, itm, n, sc, geo);
        __syncthreads();
        // ---- pairwise error
        unsigned long long bad = 0;
        for (int a = tid; a < n; a += BLOCK) {
            const T sa = sc[a];
            const double va = vv[a];
            unsigned long long cnt = 0;
            for (int b = 0; b < n; ++b) cnt += (sa >= sc[b] && va < vv[b]) ? 1u : 0u;
            bad += cnt;
        }
        const double badsum = block_sum<BLOCK>((double)bad, red);       // exact below 2^53
        const double npairs = 0.5 * (double)n * (double)(n - 1);
        // ---- top-k by score, k = min(ndcg_k, n); ties -> lower index
        const int nowk = n < ndcg_k ? n : ndcg_k;
        double dcg = 0.0;
        for (int k = 0; k < nowk; ++k) {
            T best = (T)0; int bi = -1;
            for (int p = tid; p < n; p += BLOCK) {
                const T s = sc[p];
                if (!(s != s) && (bi < 0 || s > best)) { best = s; bi = p; }       // strided scan keeps the lowest index per lane
            }
            // NaN scores (never expected) are treated as -inf: pick them last by index
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const T ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
            }
            if (BLOCK > PCR_WAVE) {
                __syncthreads();
                if (lane == 0) { wmax[wid] = best; widx[wid] = bi; }
                __syncthreads();
                best = wmax[0]; bi = widx[0];
                for (int w = 1; w < BLOCK / PCR_WAVE; ++w) {
                    const T ob = wmax[w]; const int oi = widx[w];
                    if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
                }
            }
            if (bi < 0) {                                   // only NaN scores left: take the lowest unused index
                for (int p = 0; p < n; ++p) if (sc[p] != sc[p]) { bi = p; break; }
            }
            dcg += gain[s0 + bi] * disc[k];
            __syncthreads();
            if (tid == 0) sc[bi] = -INFINITY;
            __syncthreads();
        }
        if (tid == 0) {
            out4[4 * (size_t)u] = (npairs > 0.0) ? badsum / npairs : 0.0;
            out4[4 * (size_t)u + 1] = (npairs > 0.0) ? 1.0 : 0.0;
            out4[4 * (size_t)u + 2] = dcg / idcg[u];
            out4[4 * (size_t)u + 3] = 1.0;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// k_eval2: the evaluator in O(len * T * log len) instead of O(len^2), for rating sets with at most 64
// distinct RAW rating values per user (the reference compares raw doubles, util.cpp:471-475).
// Scores are sorted by (raw level, score); then
//   #{(a,b): s_a >= s_b && v_a < v_b} = sum_a sum_{l' > l_a} #{b in run l' : s_b <= s_a}   (upper_bound)
// and the top-k by score is a k-step merge of the run tails by one wave (ties: lower index first).
// Same out4 layout as k_eval.
// ---------------------------------------------------------------------------------------
template <typename T>
static inline size_t eval2_bytes(int cap, int cap_pad, int rs_cap) {
    return carve_bytes(cap_pad, sizeof(T)) + carve_bytes(cap_pad, 4) + carve_bytes(cap, 4) + carve_bytes(rs_cap, 4);
}
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_eval2(const int64_t* __restrict__ uptr, const int32_t* __restrict__ item,
                                                 const uint16_t* __restrict__ elvl, const int64_t* __restrict__ erunofs,
                                                 const int32_t* __restrict__ erunstart, const double* __restrict__ gain,
                                                 const double* __restrict__ idcg, const double* __restrict__ disc, int ndcg_k,
                                                 const int32_t* __restrict__ users, int nusers, const T* __restrict__ U,
                                                 const T* __restrict__ Vm, Geo geo, double* __restrict__ out4, int cap, int cap_pad,
                                                 int rs_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver small(smem);
    T* vecT = small.take<T>(geo.ld);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    T* key = small.take<T>(cap_pad);
    uint32_t* li = small.take<uint32_t>(cap_pad);
    int32_t* itm = small.take<int32_t>(cap);
    int* rs = small.take<int>(rs_cap);
    const int tid = threadIdx.x;
    for (int ui = blockIdx.x; ui < nusers; ui += gridDim.x) {
        const int u = users[ui];
        const int64_t s0 = uptr[u];
        const int n = (int)(uptr[u + 1] - s0);
        const int nlev = (int)(erunofs[u + 1] - erunofs[u]) - 1;
        if (n == 0) {
            if (tid == 0) { out4[4 * (size_t)u] = 0.0; out4[4 * (size_t)u + 1] = 0.0; out4[4 * (size_t)u + 2] = 0.0; out4[4 * (size_t)u + 3] = 0.0; }
            continue;
        }
        for (int t = tid; t < geo.ld; t += BLOCK) vecT[t] = U[(size_t)u * geo.ld + t];
        for (int p = tid; p < n; p += BLOCK) itm[p] = item[s0 + p];
        for (int l = tid; l <= nlev; l += BLOCK) rs[l] = erunstart[erunofs[u] + l];
        __syncthreads();
        block_sddmm<T, BLOCK>(Vm, vecT, itm, n, key, geo);
        const int npad = next_pow2(n);
        for (int p = tid; p < npad; p += BLOCK) {
            if (p < n) li[p] = LiOps<uint32_t>::pack(elvl[s0 + p], (unsigned)p);
            else { li[p] = LiOps<uint32_t>::pack(0xFFFFu, (unsigned)p); key[p] = (T)0; }
        }
        __syncthreads();
        bitonic_sort<T, uint32_t, BLOCK, true>(key, li, npad);
        // ---- mis-ordered pairs
        double bad = 0.0;
        for (int p = tid; p < n; p += BLOCK) {
            const int lev = (int)LiOps<uint32_t>::lev(li[p]);
            const T sa = key[p];
            unsigned cnt = 0;
            for (int l = lev + 1; l < nlev; ++l) cnt += (unsigned)(ubound(key, rs[l], rs[l + 1], sa) - rs[l]);
            bad += (double)cnt;
        }
        const double badsum = block_sum<BLOCK>(bad, red);               // exact below 2^53
        const double npairs = 0.5 * (double)n * (double)(n - 1);
        // ---- top-k: wave 0 merges the run tails (lane l owns run l; nlev <= 64)
        if (tid < PCR_WAVE) {
            const int lane = tid;
            int cur = (lane < nlev) ? rs[lane + 1] - 1 : -1;
            const int lo = (lane < nlev) ? rs[lane] : 0;
            const int nowk = n < ndcg_k ? n : ndcg_k;
            double dcg = 0.0;
            for (int k = 0; k < nowk; ++k) {
                const bool have = (lane < nlev) && cur >= lo;
                T best = have ? key[cur] : (T)0;
                int bi = have ? (int)LiOps<uint32_t>::idx(li[cur]) : -1;
                int owner = lane;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const T ob = __shfl_xor(best, off);
                    const int oi = __shfl_xor(bi, off);
                    const int oo = __shfl_xor(owner, off);
                    if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; owner = oo; }
                }
                if (lane == owner) cur -= 1;
                dcg += gain[s0 + bi] * disc[k];
            }
            if (lane == 0) {
                out4[4 * (size_t)u] = (npairs > 0.0) ? badsum / npairs : 0.0;
                out4[4 * (size_t)u + 1] = (npairs > 0.0) ? 1.0 : 0.0;
                out4[4 * (size_t)u + 2] = dcg / idcg[u];
                out4[4 * (size_t)u + 3] = 1.0;
            }
        }
        __syncthreads();
    }
}

// sums of the 4 interleaved columns of out4 -> part[blk][4]; then k_fin4
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_sum4_stage1(const double* __restrict__ in, int64_t n, int per_block,
                                                               double* __restrict__ part) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = (lo + per_block < n) ? lo + per_block : n;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK)
        for (int c = 0; c < 4; ++c) a[c] += in[4 * i + c];
    for (int c = 0; c < 4; ++c) {
        const double s = block_sum<PCR_EW_BLOCK>(a[c], red);
        if (threadIdx.x == 0) part[4 * blockIdx.x + c] = s;
    }
}
// the three sums of an objective in one pass: sum(objx[0..nx)), |a|^2 over na elements, |b|^2 over nb (b may be null),
// and optionally a second per-user sum, sum(objx2[0..nx)); block-sliced partials part[blk][4] for k_fin4 (deterministic
// two-stage sums, as k_sum_stage1 / k_dots)
template <typename T>
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_obj3(const double* __restrict__ objx, const double* __restrict__ objx2, int64_t nx,
                                                        const T* __restrict__ a, int64_t na, const T* __restrict__ b, int64_t nb,
                                                        double* __restrict__ part) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    const int64_t G = gridDim.x, blk = blockIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    { const int64_t per = (nx + G - 1) / G, lo = blk * per, hi = lo + per < nx ? lo + per : nx;
      for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) s[0] += objx[i];
      if (objx2) for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) s[3] += objx2[i]; }
    { const int64_t per = (na + G - 1) / G, lo = blk * per, hi = lo + per < na ? lo + per : na;
      for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) { const double v = (double)a[i]; s[1] += v * v; } }
    if (b) { const int64_t per = (nb + G - 1) / G, lo = blk * per, hi = lo + per < nb ? lo + per : nb;
      for (int64_t i = lo + threadIdx.x; i < hi; i += PCR_EW_BLOCK) { const double v = (double)b[i]; s[2] += v * v; } }
    for (int c = 0; c < 4; ++c) {
        const double t = block_sum<PCR_EW_BLOCK>(s[c], red);
        if (threadIdx.x == 0) part[4 * blk + c] = t;
    }
}
// cnt != nullptr (the sums that follow a U step): also hands the U step's counters on -- cnt_out[0..2] = CG iterations, line
// search evaluations, cluster time-outs -- and resets the counter block (cnt[0..nzero)) for the next U step: no memset and
// no second copy on the critical path of the training loop.
__global__ __launch_bounds__(PCR_EW_BLOCK) void k_fin4(const double* __restrict__ part, int nblk, double* __restrict__ out,
                                                        unsigned long long* cnt = nullptr, double* cnt_out = nullptr, int nzero = 0,
                                                        int keep1 = 1) {
    __shared__ double red[PCR_EW_BLOCK / PCR_WAVE + 1];
    for (int c = 0; c < 4; ++c) {
        double x = 0.0;
        for (int i = threadIdx.x; i < nblk; i += PCR_EW_BLOCK) x += part[4 * i + c];
        x = block_sum<PCR_EW_BLOCK>(x, red);
        // keep1 == 0 (objective sums on ranks > 0): column 1 is the norm of a REPLICATED matrix; only rank 0 contributes it, so
        // that one all-reduce of the four columns leaves it unchanged
        if (threadIdx.x == 0) out[c] = (c == 1 && !keep1) ? 0.0 : x;
    }
    if (cnt) {
        if (threadIdx.x == 0) { cnt_out[0] = (double)cnt[0]; cnt_out[1] = (double)cnt[1]; cnt_out[2] = (double)cnt[3]; cnt_out[3] = (double)cnt[2]; }
        __syncthreads();
        for (int i = threadIdx.x; i < nzero; i += PCR_EW_BLOCK) cnt[i] = 0ull;
    }
}

// pmf-predict.cpp:58-63: pred[z] = U[user[z]] . V[item[z]]; G lanes per pair
template <typename T>
__global__ __launch_bounds__(256) void k_predict(const T* __restrict__ U, const T* __restrict__ Vm, const int32_t* __restrict__ user,
                                                 const int32_t* __restrict__ item, int64_t n, Geo geo, double* __restrict__ pred) {
    typedef typename VecOf<T>::type V;
    constexpr int VEC = VecOf<T>::N;
    const int G = geo.G, g = threadIdx.x & (G - 1);
    const int64_t z = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (z >= n) return;
    const T* up = U + (size_t)user[z] * geo.ld;
    const T* vp = Vm + (size_t)item[z] * geo.ld;
    T acc = (T)0;
    for (int ch = g; ch < geo.nchunk; ch += G)
        acc += vdot(*reinterpret_cast<const V*>(up + ch * VEC), *reinterpret_cast<const V*>(vp + ch * VEC));
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (g == 0) pred[z] = (double)acc;
}
