// pcr_plan_dev.h -- the nnz-sized part of the SpMM plan (pcr_plan.h), built on the device from the uploaded CSR.
//
// What k_spmm / k_sddmm (tile-major form) read per CSC entry -- c2r (its CSR position), crow (item), cuser (user), cuf (user | "a
// new item starts here") -- and the static slab rows of the (chunk, item) incidences (slot_base, slot_id, item_slot) used to be
// built by host threads (a counting sort per tile, three scattered 4-byte stores per rating, two more passes for the incidences)
// and uploaded: 0.75 s + 2 GB over PCIe on the Netflix shape, 16 training iterations' worth.  Here:
//     key[z] = tile(user(z)) * d2 + item[z], value = z          one wave per user (k_plan_keys)
//     stable radix sort of (key, value)                           rocPRIM (a plain library sort at set-up; not a hot-path kernel)
//     c2r / crow / cuser by gathering through the sorted values    (k_plan_gather)
//     per chunk: new-item flags, incidence count                   one wave per chunk (k_plan_flags)
//     per chunk: its incidences' items, in order                   (k_plan_inc_items)
//     stable sort of the incidences by item: the sorted position of an incidence IS its slab row (the rows of one item are
//     consecutive, in chunk order), item_slot[j] = first sorted position of item j              (k_plan_slots, k_plan_item_slot)
// The order inside a (tile, item) bucket is by user, as on the host (the sort is stable and values ascend with the user), so
// the arrays are the ones the host builder made -- checked array by array in -DPCR_PLAN_CHECK builds.
#pragma once
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

// one wave per user: key = tile * d2 + item, value = CSR position, ruser = user.  tile_u[0 .. ntiles]: first user of every tile.
template <typename K>
__global__ __launch_bounds__(256) void k_plan_keys(const int64_t* __restrict__ uptr, const int32_t* __restrict__ item,
                                                   const int64_t* __restrict__ tile_u, int ntiles, int64_t d2, int64_t nu,
                                                   K* __restrict__ key, int32_t* __restrict__ val, int32_t* __restrict__ ruser,
                                                   const unsigned char* __restrict__ excl) {
    const int lane = threadIdx.x & 63;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < nu; u += (int64_t)gridDim.x * 4) {
        int lo = 0, hi = ntiles;                       // largest t with tile_u[t] <= u (empty tiles share a boundary: the last one wins)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_u[mid] <= u) lo = mid; else hi = mid; }
        // (excl: users whose ratings the dense block kernels of pcr_vblock.h handle sort behind every tile and stay out of the chunks)
        const K base = (K)((excl && excl[u]) ? ntiles : lo) * (K)d2;
        const int64_t a = uptr[u], b = uptr[u + 1];
        for (int64_t z = a + lane; z < b; z += 64) { key[z] = base + (K)item[z]; val[z] = (int32_t)z; ruser[z] = (int32_t)u; }
    }
}

__global__ __launch_bounds__(256) void k_plan_gather(const int32_t* __restrict__ sval, const int32_t* __restrict__ item,
                                                     const int32_t* __restrict__ ruser, int32_t* __restrict__ c2r,
                                                     int32_t* __restrict__ crow, int32_t* __restrict__ cuser, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        const int32_t z = sval[p];
        c2r[p] = z; crow[p] = item[z]; cuser[p] = ruser[z];
    }
}

// out[i] = first position of the sorted keys that is not below probe[i]
template <typename K>
__global__ __launch_bounds__(256) void k_plan_lower_bounds(const K* __restrict__ skey, int64_t n, const K* __restrict__ probe, int m,
                                                           int64_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const K want = probe[i];
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (skey[mid] < want) lo = mid + 1; else hi = mid; }
    out[i] = lo;
}

// one wave per chunk: cuf = user | (new item ? sign bit : 0) -- never at a chunk's first entry -- and the chunk's incidence count
__global__ __launch_bounds__(256) void k_plan_flags(const int32_t* __restrict__ chunk_ptr, int64_t nchunks, const int32_t* __restrict__ crow,
                                                    const int32_t* __restrict__ cuser, int32_t* __restrict__ cuf, int32_t* __restrict__ inc_cnt) {
    const int lane = threadIdx.x & 63;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += (int64_t)gridDim.x * 4) {
        const int64_t a = chunk_ptr[c], b = chunk_ptr[c + 1];
        int n = 0;
        for (int64_t z0 = a; z0 < b; z0 += 64) {
            const int64_t z = z0 + lane;
            const bool f = z < b && z > a && crow[z] != crow[z - 1];
            if (z < b) cuf[z] = cuser[z] | (f ? (int32_t)0x80000000 : 0);
            n += __popcll(__ballot(f));
        }
        if (lane == 0) inc_cnt[c] = b > a ? n + 1 : 0;
    }
}

// one wave per chunk: the items of its incidences, in order, at inc_base[c]..; inc_idx = the incidence's own index (the sort's value)
__global__ __launch_bounds__(256) void k_plan_inc_items(const int32_t* __restrict__ chunk_ptr, int64_t nchunks, const int32_t* __restrict__ crow,
                                                        const int32_t* __restrict__ inc_base, int32_t* __restrict__ inc_item,
                                                        int32_t* __restrict__ inc_idx) {
    const int lane = threadIdx.x & 63;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += (int64_t)gridDim.x * 4) {
        const int64_t a = chunk_ptr[c], b = chunk_ptr[c + 1];
        int32_t o = inc_base[c];
        for (int64_t z0 = a; z0 < b; z0 += 64) {
            const int64_t z = z0 + lane;
            const bool f = z < b && (z == a || crow[z] != crow[z - 1]);
            const unsigned long long m = __ballot(f);
            if (f) { const int32_t q = o + __popcll(m & (((unsigned long long)1 << lane) - 1)); inc_item[q] = crow[z]; inc_idx[q] = q; }
            o += __popcll(m);
        }
    }
}

__global__ __launch_bounds__(256) void k_plan_slots(const int32_t* __restrict__ sidx, int32_t* __restrict__ slot_id, int64_t n) {
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n; s += (int64_t)gridDim.x * 256) slot_id[sidx[s]] = (int32_t)s;
}
// item_slot[j] = first sorted incidence whose item is not below j, j = 0 .. d2 (item_slot[d2] = all of them)
__global__ __launch_bounds__(256) void k_plan_item_slot(const int32_t* __restrict__ sitem, int64_t n, int64_t d2, int32_t* __restrict__ item_slot) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j <= d2; j += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sitem[mid] < j) lo = mid + 1; else hi = mid; }
        item_slot[j] = (int32_t)lo;
    }
}

static inline int plan_bits(unsigned long long max_key) { int b = 1; while (b < 64 && (max_key >> b) != 0) ++b; return b; }

// stable sort of n (key, value) pairs by the low `bits` bits of the key; all four arrays on the device
template <typename K>
static inline hipError_t plan_sort_pairs(const K* kin, K* kout, const int32_t* vin, int32_t* vout, size_t n, int bits, hipStream_t st) {
    if (n == 0) return hipSuccess;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, st);
    if (e != hipSuccess) return e;
    void* tmp = nullptr;
    e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 1));
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)bits, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    return e != hipSuccess ? e : e2;
}
