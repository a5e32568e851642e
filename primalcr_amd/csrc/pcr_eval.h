// pcr_eval.h -- evaluator (compute_pairwise_error_ndcg, util.cpp:434-542), the objective's reduction kernels, k_predict
// (pmf-predict.cpp:56-64).  Part of pcr_kernels.h.
#pragma once
#include "pcr_prims.h"
#include "pcr_vside.h"

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
// BIG (users beyond 4096 ratings): the n-sized arrays live in a per-workgroup global scratch slice, (level, index) packs into 64 bits
template <typename T>
static inline size_t eval2_big_bytes(int cap, int cap_pad) {
    return carve_bytes(cap_pad, sizeof(T)) + carve_bytes(cap_pad, 8) + carve_bytes(cap, 4);
}
template <typename T, int BLOCK, bool BIG = false>
__global__ __launch_bounds__(BLOCK) void k_eval2(const int64_t* __restrict__ uptr, const int32_t* __restrict__ item,
                                                 const uint16_t* __restrict__ elvl, const int64_t* __restrict__ erunofs,
                                                 const int32_t* __restrict__ erunstart, const double* __restrict__ gain,
                                                 const double* __restrict__ idcg, const double* __restrict__ disc, int ndcg_k,
                                                 const int32_t* __restrict__ users, int nusers, const T* __restrict__ U,
                                                 const T* __restrict__ Vm, Geo geo, double* __restrict__ out4, int cap, int cap_pad,
                                                 int rs_cap, char* scratch = nullptr, size_t stride = 0) {
    typedef typename LiSel<T, BIG>::type LI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Carver small(smem);
    T* vecT = small.take<T>(geo.ld);
    double* red = small.take<double>(BLOCK / PCR_WAVE + 1);
    int* rs = small.take<int>(rs_cap);
    Carver big(BIG ? scratch + (size_t)blockIdx.x * stride : small.p);
    T* key = big.take<T>(cap_pad);
    LI* li = big.take<LI>(cap_pad);
    int32_t* itm = big.take<int32_t>(cap);
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
            if (p < n) li[p] = LiOps<LI>::pack(elvl[s0 + p], (unsigned)p);
            else { li[p] = LiOps<LI>::pack(0xFFFFu, (unsigned)p); key[p] = (T)0; }
        }
        __syncthreads();
        bitonic_sort<T, LI, BLOCK, true, !BIG>(key, li, npad, n);
        // ---- mis-ordered pairs
        double bad = 0.0;
        for (int p = tid; p < n; p += BLOCK) {
            const int lev = (int)LiOps<LI>::lev(li[p]);
            const T sa = key[p];
            unsigned long long cnt = 0;
            for (int l = lev + 1; l < nlev; ++l) cnt += (unsigned long long)(ubound(key, rs[l], rs[l + 1], sa) - rs[l]);
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
                int bi = have ? (int)LiOps<LI>::idx(li[cur]) : -1;
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
// fp64 row-major rows x r (the reference's mat_t payload) <-> the device's rows x ld matrix of T (pad columns zero)
template <typename T>
__global__ __launch_bounds__(256) void k_mat_in(const double* __restrict__ src, T* __restrict__ dst, int64_t rows, int r, int ld) {
    const int64_t n = rows * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / ld;
        const int col = (int)(i - row * ld);
        dst[i] = col < r ? (T)src[row * r + col] : (T)0;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_mat_out(const T* __restrict__ src, double* __restrict__ dst, int64_t rows, int r, int ld) {
    const int64_t n = rows * r;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / r;
        dst[i] = (double)src[row * ld + (i - row * r)];
    }
}

// gain[z] = gain of the rating's level (util.cpp:519: pow(2, v) - 1, evaluated once per (user, level) on the host and looked up
// here): one wave per user, coalesced over its ratings.  lgain uses the slots of run_start (run_ofs[u] + level).
__global__ __launch_bounds__(256) void k_gain_from_levels(const int64_t* __restrict__ uptr, const uint16_t* __restrict__ lvl,
                                                          const int64_t* __restrict__ runofs, const double* __restrict__ lgain,
                                                          double* __restrict__ gain, int64_t nu) {
    const int lane = threadIdx.x & 63;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < nu; u += (int64_t)gridDim.x * 4) {
        const int64_t a = uptr[u], b = uptr[u + 1];
        const double* g = lgain + runofs[u];
        for (int64_t z = a + lane; z < b; z += 64) gain[z] = g[lvl[z]];
    }
}

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

