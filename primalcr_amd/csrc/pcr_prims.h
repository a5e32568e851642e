// pcr_prims.h -- wave / team primitives of the PrimalCR++ kernels (gfx950, wave64): cross-lane exchanges, scans, the
// (level, m) sort (nearly-sorted fast path + bitonic network), window search, sweep coefficients, per-user row primitives
// (block_sddmm, block_gather_axpy, LDS row images).  Part of pcr_kernels.h (kernel map and formulation there).
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
    // k_prepare's per-user back-off: 1 = the user's last fast-path attempt failed (both tiers, then the network anyway: the
    // long users of the first ~10 iterations) -> the next prepare sorts it with the network straight away and sets 2 = try again
    unsigned char* rhint;
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

// ascending bitonic sort of (key, li)[0, n) by (level, key); npad = pow2 >= n.  The network is the NORMALISED one: every
// compare-exchange orders its pair ascending (the first stage of every merge pairs i with its mirror image in the 2 j-block
// instead of reversing the direction of every other block), so the elements [n, npad) are VIRTUAL -- "+infinity" that no exchange
// would ever move: a pair that reaches beyond n is skipped, and the arrays need n entries, not npad (the longest user of the
// headline shape has 2634 ratings: 42 instead of 54 KB of LDS per workgroup of k_prepare_all, three workgroups per CU instead
// of two).  Tie order among equal (level, key) is irrelevant to every sum computed from the order (the reference's std::sort
// is unstable too).
// TIE = true additionally orders equal (level, key) by DESCENDING index (k_eval2: the lowest index then
// sits at the end of its run and is taken first).
// Compare-exchange network over LDS.  With stride j <= 64 the pairs a wave works on (64 consecutive t) lie in ITS OWN
// aligned 128-element chunk, in every such stage alike (the mirror pairs of a merge of 2 j <= 128 elements too), so between two
// short-stride stages a wave-level ordering point replaces the workgroup barrier: of the 78 stages of a 4096-element sort only
// 21 need __syncthreads().
// (INLDS = false: the arrays live in global scratch, every stage keeps the workgroup barrier.)
template <typename T, typename LI, int BLOCK, bool TIE = false, bool INLDS = true>
__device__ __forceinline__ void bitonic_sort(T* key, LI* li, int npad, int n) {
    const int tid = btid<BLOCK>();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (npad >> 1); t += BLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i | j);
                if (l >= n) continue;                                 // (i < l: the pair's upper element is virtual)
                T ka = key[i], kb = key[l];
                LI la = li[i], lb = li[l];
                unsigned va = LiOps<LI>::lev(la), vb = LiOps<LI>::lev(lb);
                bool b_lt_a = (vb < va) || (vb == va && kb < ka);
                if (TIE && va == vb && ka == kb) b_lt_a = LiOps<LI>::idx(lb) > LiOps<LI>::idx(la);
                if (b_lt_a) { key[i] = kb; key[l] = ka; li[i] = lb; li[l] = la; }
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

