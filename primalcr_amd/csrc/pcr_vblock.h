// pcr_vblock.h -- the BLOCKED-USER V step on the matrix cores (second object; the north star's "MFMA only for the
// batched rank-k GEMM-like V-update step").  Optional (pcr_tune "vblock_users"): the default V step gathers one factor row per
// rating for every user (k_sddmm / k_spmm, pcr_vside.h).
//
// The users with the most ratings rate a large share of the item catalogue (ml1m: the 64 longest users hold 1000-2600 of 3952
// items), so their slice of the two rating-parallel products of a Hessian-vector product (compute_Ha_new, pcrpp.cpp:252-332) is
// close to dense:
//     b_ij  = u_i . a_j            for the block's ratings      =  the rated entries of  U_B A^T      (B users x d2 items)
//     Hp_j += sum_i c_ij u_i       over the block's users       =  C_B^T U_B,  C_B the block's coefficients as a dense B x d2 matrix
// Both are GEMMs with K = the rank resp. K = the block's user count, run here on v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64
// with a STATIC dense index (cpos[i][j] = CSR position of user i's rating of item j, or -1) as the epilogue's / prologue's
// gather map; the block's ratings are left out of the tile-major CSC the sparse kernels walk (pcr_plan_dev.h).  b is written in
// CSR order (the sweeps pick it up through sidx, as after the CSC-form SDDMM), c is read in CSR order.
#pragma once
#include "pcr_kernels.h"
#include "pcr_gram.h"

// b[cpos[i][j]] = U[blk_user[i]] . A[j] for every rated (i, j) of the block: one wave per (TS users) x (TS items) tile
template <typename T>
__global__ __launch_bounds__(256) void k_vblock_b(const T* __restrict__ U, const T* __restrict__ A, const int32_t* __restrict__ blk_user,
                                                  int nbp, const int32_t* __restrict__ cpos, int64_t d2, Geo geo, T* __restrict__ out,
                                                  const int* skip) {
    typedef GramMfma<T> MM;
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t jt = ((int64_t)d2 + MM::TS - 1) / MM::TS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wid;
    const int it = nbp / MM::TS;
    if (tile >= (int64_t)it * jt) return;
    const int I = (int)(tile % it);
    const int64_t J = tile / it;
    const int ld = geo.ld, row = lane % MM::TS, kh = lane / MM::TS, Lk = ld / MM::KS;
    const int bu = blk_user[I * MM::TS + row];
    const int64_t j = J * MM::TS + row;
    const T* ua = U + (size_t)(bu >= 0 ? bu : 0) * ld + kh * Lk;
    const T* ab = A + (size_t)(j < d2 ? j : 0) * ld + kh * Lk;
    const T za = bu >= 0 ? (T)1 : (T)0, zb = j < d2 ? (T)1 : (T)0;
    typename MM::acc_t acc;
#pragma unroll
    for (int e = 0; e < MM::NACC; ++e) acc[e] = (T)0;
    // (eight operand pairs in flight: the k loop is a chain of dependent global loads otherwise -- 25 us per launch on the ml1m shape)
    int s = 0;
    for (; s + 8 <= Lk; s += 8) {
        T a8[8], b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a8[e] = ua[s + e]; b8[e] = ab[s + e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = MM::mma(a8[e] * za, b8[e] * zb, acc);
    }
    for (; s < Lk; ++s) acc = MM::mma(ua[s] * za, ab[s] * zb, acc);
    const int64_t col = J * MM::TS + MM::ccol(lane);
#pragma unroll
    for (int e = 0; e < MM::NACC; ++e) {
        const int i = I * MM::TS + MM::crow(e, lane);
        if (i < nbp && col < d2) {
            const int32_t pos = cpos[(size_t)i * d2 + col];
            if (pos >= 0) out[pos] = acc[e];
        }
    }
}

// out[j][:] += sum_i c[cpos[i][j]] U[blk_user[i]][:]: one wave per (TS items) x (TS columns) tile, K = the block's users
template <typename T>
__global__ __launch_bounds__(256) void k_vblock_hp(const T* __restrict__ U, const T* __restrict__ c, const int32_t* __restrict__ blk_user,
                                                   int nbp, const int32_t* __restrict__ cpos, int64_t d2, Geo geo, T* __restrict__ out,
                                                   const int* skip) {
    typedef GramMfma<T> MM;
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ld = geo.ld;
    const int ct = (ld + MM::TS - 1) / MM::TS;
    const int64_t jt = ((int64_t)d2 + MM::TS - 1) / MM::TS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wid;
    if (tile >= jt * ct) return;
    const int C = (int)(tile % ct);
    const int64_t J = tile / ct;
    const int row = lane % MM::TS, kh = lane / MM::TS, Lk = nbp / MM::KS;
    const int64_t j = J * MM::TS + row;                          // A operand: item j of the tile (M index), users kh * Lk .. (K)
    const int colb = C * MM::TS + row;                           // B operand: column colb of U (N index), the same users
    typename MM::acc_t acc;
#pragma unroll
    for (int e = 0; e < MM::NACC; ++e) acc[e] = (T)0;
    // (four users in flight: user id -> position -> coefficient is a chain of three dependent loads per step)
    for (int s0 = 0; s0 < Lk; s0 += 4) {
        int bu4[4];
        int32_t pos4[4];
        T a4[4], b4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bu4[e] = s0 + e < Lk ? blk_user[kh * Lk + s0 + e] : -1;
#pragma unroll
        for (int e = 0; e < 4; ++e) pos4[e] = (bu4[e] >= 0 && j < d2) ? cpos[(size_t)(kh * Lk + s0 + e) * d2 + j] : -1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a4[e] = pos4[e] >= 0 ? c[pos4[e]] : (T)0;
            b4[e] = (bu4[e] >= 0 && colb < ld) ? U[(size_t)bu4[e] * ld + colb] : (T)0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) if (s0 + e < Lk) acc = MM::mma(a4[e], b4[e], acc);
    }
    const int col = C * MM::TS + MM::ccol(lane);
#pragma unroll
    for (int e = 0; e < MM::NACC; ++e) {
        const int64_t jr = J * MM::TS + MM::crow(e, lane);
        if (jr < d2 && col < ld) out[(size_t)jr * ld + col] += acc[e];
    }
}
