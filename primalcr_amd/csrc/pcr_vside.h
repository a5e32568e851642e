// pcr_vside.h -- kernels of the V step (update_V_new, pcrpp.cpp:415-444): k_sddmm, k_prepare, k_vsweep*, k_spmm + k_spmm_fin,
// the CG vector kernels.  Part of pcr_kernels.h.
#pragma once
#include "pcr_prims.h"

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
    // (cap_pad: what the caller sizes the sort arrays with -- cap is enough, the sort pads virtually)
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
        const bool may_prev = !BIG && sizeof(T) == 4 && S.resort_d > 0 && S.prev_valid;
        // (a user whose last attempt failed both tiers -- the long users of the first ~10 iterations, whose order still moves -- is
        // sorted by the network straight away this time, from the CSR order, and tries again next time: a failed attempt costs
        // ~110 LDS operations per rating on top of the network.  ml1m: 1.432 -> 1.425 ms per step, inside the noise; kept for the
        // shapes whose long users never settle)
        const bool from_prev = may_prev && S.rhint[u] != 1;
#pragma unroll 4
        for (int p = tid; p < n; p += BLOCK) {                       // (no padding: the sort's elements beyond n are virtual)
            if (from_prev) { const unsigned idx = (unsigned)S.sidx[s0 + p]; li[p] = LiOps<LI>::pack(S.slvl[s0 + p], idx); key[p] = m_in[s0 + idx]; }
            else { li[p] = LiOps<LI>::pack(S.lvl[s0 + p], (unsigned)p); key[p] = m_in[s0 + p]; }
        }
        bsync<BLOCK>();
        PPROF(0);
        // (tmp and the second key array share the prefix-sum array, idle until the loss: 4-byte scores only.  With 8-byte scores
        // there is room for tmp + a permuted index array but not for a second key array; that variant -- keys verified through
        // tmp, the scores re-read from global memory in their new order -- was built and measured SLOWER than the network
        // (one-wave teams 22.6 -> 41.5 kclk per user, 512-thread teams 67.5 -> 75.9: the re-read is one more dependent global
        // round trip in a chain of ~10 us; NOTES.md round 4), so fp64 sorts fully here)
        bool resorted = false;
        if constexpr (!BIG && sizeof(T) == 4)
            if (from_prev)
                resorted = resort_window<T, LI, BLOCK>(key, li, [&](int p) { return (int)LiOps<LI>::lev(li[p]); }, rs, n, reinterpret_cast<int*>(Sx),
                                                       reinterpret_cast<T*>(reinterpret_cast<int*>(Sx) + n), S.resort_d, reinterpret_cast<int*>(red));
        if (!resorted) bitonic_sort<T, LI, BLOCK, false, !BIG>(key, li, npad, n);
        if (may_prev && tid == 0) S.rhint[u] = from_prev ? (resorted ? 0 : 1) : 2;
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
// MINW: the minimum-waves-per-SIMD bound of the symbol (8: at most 64 VGPRs -- four 512-thread workgroups per CU instead of three --
// which also caps the launch's dynamic LDS at a quarter of a CU's: for shards whose longest users fit 40 KB; 1: no bound)
template <typename T, bool HV, int WB, int MINW = 1>
__global__ __launch_bounds__(WB, MINW) void k_vsweep_all(Shard<T> S, const int32_t* __restrict__ users_a, int nusers_a, int cap_a,
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

