// pcr_plan.h -- host side of the plan of the item-major SpMM (k_spmm / k_spmm_fin, pcr_vside.h): the user tiles, the chunk lists,
// the item ranges and the workgroup -> chunk map that binds a user tile to its XCD(s) -- everything that follows from the row
// pointers and a few cut positions.  The nnz-sized part (the tile-major CSC itself, the new-item flags, the (chunk, item)
// incidences with their static item-major slab rows) is built ON THE DEVICE from the uploaded CSR: pcr_plan_dev.h.  No GPU calls here.
// Replaces the scatter of pcrpp.cpp:240-243 / :323-327 (one `omp atomic` per scalar) by a static plan built once per data set.
#pragma once
#include <cmath>
#include <algorithm>
#include <cstdint>
#include <vector>

#include "pcr_host.h"

struct SpmmPlanIn {
    const std::vector<int64_t>& uptr;     // user-major CSR of the shard
    const int32_t* item;                  // (the data set's own array: no copy)
    int64_t nu, nnz, d2;
    int ld, G, ncu;                       // padded row length, lanes per row, CUs of the device
    size_t esz;                           // bytes per stored element
    int tune_chunk, tune_tiles, tune_ranges;   // pcr_tune overrides (0: choose)
};
struct SpmmPlan {
    int chunk = 128, ntiles = 1, n_rng = 1, blocks = 0;
    std::vector<int64_t> rng_item;        // item range r = [rng_item[r], rng_item[r + 1])
    std::vector<int> rng_blk;             // its workgroups: blk[rng_blk[r] .. rng_blk[r + 1])
    std::vector<int32_t> cpos, cuser, crow, ruser, chunk_ptr, inc_base, slot_id, item_slot, cuf;
    std::vector<int2> blk;
    size_t slab_rows = 0;
};

// ---- the plan in three host steps around the device build (pcr_plan_dev.h) -------------------------------------------------------

// chunk = ratings one lane group of k_spmm walks (one slab row per item it meets).  A shard whose lane groups all fit on the chip at
// once (6 workgroups per CU at 80 VGPRs) gets the smallest chunk that still fits in ONE round -- more, shorter chains (ml1m fp32: 96
// instead of 128, k_spmm 32.3 -> 30.3 us; 64: 33.3 us, a second round; 192: 40.3 us).  A shard that needs between one and four rounds
// at 128 AND whose lane groups are whole waves (G = 64: fp64 at k = 100, wide ranks) takes, of 128 / 96 / 64, the chunk whose LAST round
// is fullest (round 6: ml1m in fp64 needs 1.19 rounds at 128 -- a second round for a fifth of the work: k_spmm 50.8 us -- against 1.59
// at 96: 47.0 us, the fp64 step 2.40 -> 2.345 ms).  Everything else: 128 (large shards: the tail is one round in dozens).
static inline int plan_pick_chunk(int64_t nnz_local, int ncu, int G) {
    const int64_t groups_at_once = (int64_t)ncu * 6 * (256 / G);
    const int64_t fit = (std::max<int64_t>(nnz_local, 1) + groups_at_once - 1) / groups_at_once;
    if (fit <= 128) return (int)std::max<int64_t>(64, (fit + 31) / 32 * 32);
    if (fit >= 4 * 128 || G < 64) return 128;                    // (32-lane groups: a 2 M-rating fp32 shard measured 1.2 % SLOWER under the rule below)
    int best = 128;
    double best_idle = 2.0;
    for (int c : {128, 96, 64}) {
        const double rounds = (double)fit / c, full = std::ceil(rounds);
        const double idle = (full - rounds) / full;               // share of the slots the last round leaves empty
        if (idle < best_idle - 1e-9) { best_idle = idle; best = c; }      // (ties: the larger chunk, fewer slab rows)
    }
    return best;
}
// 1. plan_tiles: chunk length, user tiles (tile_u), item ranges -- from the row pointers alone
static inline void plan_tiles(const SpmmPlanIn& in, SpmmPlan& P, std::vector<int64_t>& tile_u) {
    const std::vector<int64_t>& uptr = in.uptr;
    const int64_t nu = in.nu, nnz_local = in.nnz, d2 = in.d2;
    const struct { int ld, G; } geo{in.ld, in.G};
    const int ncu = in.ncu;
    const struct { int spmm_chunk, spmm_tiles, allreduce_chunks; } tune{in.tune_chunk, in.tune_tiles, in.tune_ranges};
    int& spmm_chunk = P.chunk;
    int& n_rng = P.n_rng;
    auto& rng_item = P.rng_item;
    // Tile-major CSC of the shard (pcr_kernels.h, k_spmm): users are cut into tiles of about equal rating count whose
    // rows of U take at most 1.25 MB (measured on a 48 k x 17.8 k, 10 M shape: 1.2 MB tiles 379 us, 2.4 MB tiles 595 us =
    // untiled, 0.6 MB tiles 411 us: the tile shares the XCD's 4 MB L2 with the streamed ids, c and the slab stores);
    // inside a tile the entries are ordered by item, then user.
    {
        spmm_chunk = plan_pick_chunk(nnz_local, ncu, geo.G);     // ratings one lane group walks
        if (tune.spmm_chunk > 0) spmm_chunk = std::max(8, tune.spmm_chunk);
        const size_t row_bytes = (size_t)geo.ld * in.esz;
        int64_t tile_users_max = std::max<int64_t>(64, (int64_t)((5u << 18) / row_bytes));
        // Small shards: what a tile re-reads is its rows of U and its slice of c (4-byte gathers through the static map); 2, 4 or 8
        // tiles -- each bound to 4, 2 or 1 XCDs -- whichever is the fewest that keeps that under 2 MB per tile.  Every (tile, item)
        // pair costs a slab row, so fewer tiles is less slab: ml1m 4 tiles, 25 k rows = 10 MB written by k_spmm and read back by
        // k_spmm_fin instead of 41 k rows = 16 MB at 8 tiles (k_spmm_fin 9.6 -> 8.3 us, k_spmm unchanged; 2 tiles: 7.8 / +0.7 us).
        int64_t small_tiles = std::min<int64_t>(8, nu / 256);
        if (small_tiles >= 2) {
            const double reread = (double)nu * row_bytes + (double)nnz_local * in.esz;
            small_tiles = reread / 2 <= 2e6 ? 2 : reread / 4 <= 2e6 ? 4 : 8;
            small_tiles = std::min<int64_t>(small_tiles, nu / 256 >= 8 ? 8 : nu / 256 >= 4 ? 4 : 2);
        }
        int64_t ntiles = std::max<int64_t>(cdiv(nu, tile_users_max), small_tiles);
        // ... but every (tile, item) pair with a rating costs a partial row in the slab: on a very wide, sparse item side
        // (Yahoo-shaped: 136 k items) 1.25 MB tiles would hold ~5 ratings per pair and the slab would outweigh the
        // gather.  Keep at least 16 ratings per pair on average (ml1m 32, Netflix shape 37: unaffected).
        ntiles = std::min<int64_t>(ntiles, std::max<int64_t>(8, nnz_local / (16 * std::max<int64_t>(d2, 1))));
        if (ntiles > 4) ntiles = (ntiles + 7) / 8 * 8;             // every XCD the same number of tiles (2 and 4 tiles: XCD groups)
        else if (ntiles == 3) ntiles = 4;
        if (tune.spmm_tiles > 0) ntiles = tune.spmm_tiles;
        ntiles = std::max<int64_t>(1, std::min<int64_t>(ntiles, std::max<int64_t>(nu, 1)));
        tile_users_max = std::max<int64_t>(tile_users_max, 2 * (int64_t)cdiv(nu, ntiles));       // (the density bound may ask for larger tiles)
        tile_u.assign(1, 0);                                     // user boundaries: equal ratings, at most tile_users_max users
        for (int64_t t = 1; t < ntiles; ++t) {
            const int64_t want = nnz_local * t / ntiles;
            int64_t u = std::lower_bound(uptr.begin(), uptr.end(), want) - uptr.begin();
            u = std::min(u, tile_u.back() + tile_users_max);
            u = std::max(u, tile_u.back());
            u = std::min<int64_t>(u, nu);
            tile_u.push_back(u);
        }
        tile_u.push_back(nu);
        while ((int64_t)tile_u.size() >= 2 && nu - tile_u[tile_u.size() - 2] > tile_users_max) {     // the cap pushed users to the end
            tile_u.back() = tile_u[tile_u.size() - 2] + tile_users_max;
            tile_u.push_back(nu);
        }
        ntiles = (int64_t)tile_u.size() - 1;
        P.ntiles = (int)ntiles;
        // item ranges (see n_rng): OPT-IN through pcr_tune("allreduce_chunks", n) -- the overlap of one range's all-reduce with the
        // next range's SpMM is equality-tested with several ranks on one device, but has never run across two physical GPUs
        // (no multi-GPU node was available to this build), so the default exchange is the plain one: one all-reduce per vector
        // on the solver's stream.  Where it should pay: vectors of 16 MB and more (the Yahoo!Music shape's 109 MB: one exchange
        // ~ a third of a CG iteration at N = 8), about one range per 4 MB, at most 8; a range costs two more launches and its
        // own ramp and tail, which an exchange of a few MB does not pay for.
        n_rng = 1;
        if (tune.allreduce_chunks > 0) n_rng = tune.allreduce_chunks;
        n_rng = (int)std::max<int64_t>(1, std::min<int64_t>(n_rng, std::min<int64_t>(64, d2)));
        rng_item.assign(n_rng + 1, 0);
        for (int r = 0; r <= n_rng; ++r) rng_item[r] = d2 * r / n_rng;
    }
}

// 2. plan_chunks: the chunk list of every (tile, item range) from where the tile's sorted entries cross into each range
//    (cut[t * (n_rng + 1) + r], from the device's sorted keys; cut[.. + 0] / [.. + n_rng] = the tile's first / past-the-end entry).
//    Chunks never straddle tiles or ranges.  trc0[t * n_rng + r] = first chunk of (tile, range).
static inline void plan_chunks(SpmmPlan& P, int64_t nnz_local, const std::vector<int64_t>& cut, std::vector<int32_t>& trc0) {
    const int64_t ntiles = P.ntiles;
    const int n_rng = P.n_rng;
    trc0.assign((size_t)ntiles * n_rng + 1, 0);
    P.chunk_ptr.clear();
    for (int64_t t = 0; t < ntiles; ++t)
        for (int r = 0; r < n_rng; ++r) {
            trc0[(size_t)t * n_rng + r] = (int32_t)P.chunk_ptr.size();
            for (int64_t a = cut[(size_t)t * (n_rng + 1) + r]; a < cut[(size_t)t * (n_rng + 1) + r + 1]; a += P.chunk) P.chunk_ptr.push_back((int32_t)a);
        }
    trc0[(size_t)ntiles * n_rng] = (int32_t)P.chunk_ptr.size();
    P.chunk_ptr.push_back((int32_t)nnz_local);                 // (the end of the last chunk: the entries the chunks cover)
}

// 3. plan_blocks: workgroup -> chunks: gpb chunks per workgroup, never across tiles; tile t is walked by workgroups b = t mod 8 (mod 8).
//    One plan per item range, back to back (each a multiple of 8 workgroups, so the affinity holds in a launch of one range
//    as in a launch of all of them).
static inline void plan_blocks(SpmmPlan& P, int G, const std::vector<int32_t>& trc0) {
    const int64_t ntiles = P.ntiles;
    const int n_rng = P.n_rng;
    auto& blk = P.blk; auto& rng_blk = P.rng_blk;
    const int gpb = 256 / G;
    blk.clear();
    rng_blk.assign(n_rng + 1, 0);
    for (int r = 0; r < n_rng; ++r) {
        std::vector<std::vector<int2>> per_xcd(8);
        size_t rr8 = 0;
        for (int64_t t = 0; t < ntiles; ++t) {
            const int32_t c0 = trc0[(size_t)t * n_rng + r], c1 = trc0[(size_t)t * n_rng + r + 1];
            // tile t -> XCD t mod 8; 2 or 4 tiles: tile t -> the XCDs {t, t + ntiles, ..} in turn (each of them then caches only
            // that tile's rows of U and slice of c); any other count below 8: no affinity, use every XCD
            size_t turn = 0;
            for (int32_t c = c0; c < c1; c += gpb) {
                const size_t x = ntiles >= 8 ? (size_t)(t % 8) : (ntiles == 2 || ntiles == 4) ? (size_t)t + (size_t)ntiles * (turn++ % (8 / ntiles)) : rr8++ % 8;
                per_xcd[x].push_back(make_int2(c, std::min<int32_t>(gpb, c1 - c)));
            }
        }
        size_t deepest = 0;
        for (auto& v : per_xcd) deepest = std::max(deepest, v.size());
        const size_t base = blk.size();
        blk.resize(base + deepest * 8, make_int2(0, 0));
        for (int x = 0; x < 8; ++x)
            for (size_t i = 0; i < per_xcd[x].size(); ++i) blk[base + i * 8 + x] = per_xcd[x][i];
        rng_blk[r + 1] = (int)blk.size();
    }
    if (blk.empty()) blk.push_back(make_int2(0, 0));
    P.blocks = rng_blk[n_rng];
}

#ifdef PCR_PLAN_CHECK
// The whole plan on the host, as rounds 1-4 built it: kept for -DPCR_PLAN_CHECK builds, where pcr_solver_create builds both and
// compares every array (developer check of pcr_plan_dev.h; never in the shipped library).
static inline void build_spmm_plan_host(const SpmmPlanIn& in, SpmmPlan& P) {
    const std::vector<int64_t>& uptr = in.uptr;
    const int32_t* item = in.item;
    const int64_t nu = in.nu, nnz_local = in.nnz, d2 = in.d2;
    const struct { int ld, G; } geo{in.ld, in.G};
    const int ncu = in.ncu;
    const struct { int spmm_chunk, spmm_tiles, allreduce_chunks; } tune{in.tune_chunk, in.tune_tiles, in.tune_ranges};
    int& spmm_chunk = P.chunk;
    int& n_rng = P.n_rng;
    auto &rng_item = P.rng_item; auto &rng_blk = P.rng_blk;
    auto &cpos = P.cpos, &cuser = P.cuser, &crow = P.crow, &ruser = P.ruser, &chunk_ptr = P.chunk_ptr, &inc_base = P.inc_base,
         &slot_id = P.slot_id, &item_slot = P.item_slot, &cuf = P.cuf;
    auto& blk = P.blk;
    // Tile-major CSC of the shard (pcr_kernels.h, k_spmm): users are cut into tiles of about equal rating count whose
    // rows of U take at most 1.25 MB (measured on a 48 k x 17.8 k, 10 M shape: 1.2 MB tiles 379 us, 2.4 MB tiles 595 us =
    // untiled, 0.6 MB tiles 411 us: the tile shares the XCD's 4 MB L2 with the streamed ids, c and the slab stores);
    // inside a tile the entries are ordered by item, then user.
    cpos.resize(nnz_local); cuser.resize(nnz_local); crow.resize(nnz_local); ruser.resize(nnz_local);
    pcr_parallel_ranges(nu, pcr_host_threads(), [&](int, int64_t lo, int64_t hi) {
        for (int64_t u = lo; u < hi; ++u)
            for (int64_t z = uptr[u]; z < uptr[u + 1]; ++z) ruser[z] = (int32_t)u;
    });
    {
        spmm_chunk = plan_pick_chunk(nnz_local, ncu, geo.G);     // ratings one lane group walks
        if (tune.spmm_chunk > 0) spmm_chunk = std::max(8, tune.spmm_chunk);
        const size_t row_bytes = (size_t)geo.ld * in.esz;
        int64_t tile_users_max = std::max<int64_t>(64, (int64_t)((5u << 18) / row_bytes));
        // Small shards: what a tile re-reads is its rows of U and its slice of c (4-byte gathers through the static map); 2, 4 or 8
        // tiles -- each bound to 4, 2 or 1 XCDs -- whichever is the fewest that keeps that under 2 MB per tile.  Every (tile, item)
        // pair costs a slab row, so fewer tiles is less slab: ml1m 4 tiles, 25 k rows = 10 MB written by k_spmm and read back by
        // k_spmm_fin instead of 41 k rows = 16 MB at 8 tiles (k_spmm_fin 9.6 -> 8.3 us, k_spmm unchanged; 2 tiles: 7.8 / +0.7 us).
        int64_t small_tiles = std::min<int64_t>(8, nu / 256);
        if (small_tiles >= 2) {
            const double reread = (double)nu * row_bytes + (double)nnz_local * in.esz;
            small_tiles = reread / 2 <= 2e6 ? 2 : reread / 4 <= 2e6 ? 4 : 8;
            small_tiles = std::min<int64_t>(small_tiles, nu / 256 >= 8 ? 8 : nu / 256 >= 4 ? 4 : 2);
        }
        int64_t ntiles = std::max<int64_t>(cdiv(nu, tile_users_max), small_tiles);
        // ... but every (tile, item) pair with a rating costs a partial row in the slab: on a very wide, sparse item side
        // (Yahoo-shaped: 136 k items) 1.25 MB tiles would hold ~5 ratings per pair and the slab would outweigh the
        // gather.  Keep at least 16 ratings per pair on average (ml1m 32, Netflix shape 37: unaffected).
        ntiles = std::min<int64_t>(ntiles, std::max<int64_t>(8, nnz_local / (16 * std::max<int64_t>(d2, 1))));
        if (ntiles > 4) ntiles = (ntiles + 7) / 8 * 8;             // every XCD the same number of tiles (2 and 4 tiles: XCD groups)
        else if (ntiles == 3) ntiles = 4;
        if (tune.spmm_tiles > 0) ntiles = tune.spmm_tiles;
        ntiles = std::max<int64_t>(1, std::min<int64_t>(ntiles, std::max<int64_t>(nu, 1)));
        tile_users_max = std::max<int64_t>(tile_users_max, 2 * (int64_t)cdiv(nu, ntiles));       // (the density bound may ask for larger tiles)
        std::vector<int64_t> tile_u(1, 0);                       // user boundaries: equal ratings, at most tile_users_max users
        for (int64_t t = 1; t < ntiles; ++t) {
            const int64_t want = nnz_local * t / ntiles;
            int64_t u = std::lower_bound(uptr.begin(), uptr.end(), want) - uptr.begin();
            u = std::min(u, tile_u.back() + tile_users_max);
            u = std::max(u, tile_u.back());
            u = std::min<int64_t>(u, nu);
            tile_u.push_back(u);
        }
        tile_u.push_back(nu);
        while ((int64_t)tile_u.size() >= 2 && nu - tile_u[tile_u.size() - 2] > tile_users_max) {     // the cap pushed users to the end
            tile_u.back() = tile_u[tile_u.size() - 2] + tile_users_max;
            tile_u.push_back(nu);
        }
        ntiles = (int64_t)tile_u.size() - 1;
        // item ranges (see n_rng): OPT-IN through pcr_tune("allreduce_chunks", n) -- the overlap of one range's all-reduce with the
        // next range's SpMM is equality-tested with several ranks on one device, but has never run across two physical GPUs
        // (no multi-GPU node was available to this build), so the default exchange is the plain one: one all-reduce per vector
        // on the solver's stream.  Where it should pay: vectors of 16 MB and more (the Yahoo!Music shape's 109 MB: one exchange
        // ~ a third of a CG iteration at N = 8), about one range per 4 MB, at most 8; a range costs two more launches and its
        // own ramp and tail, which an exchange of a few MB does not pay for.
        {
            n_rng = 1;
            if (tune.allreduce_chunks > 0) n_rng = tune.allreduce_chunks;
            n_rng = (int)std::max<int64_t>(1, std::min<int64_t>(n_rng, std::min<int64_t>(64, d2)));
            rng_item.assign(n_rng + 1, 0);
            for (int r = 0; r <= n_rng; ++r) rng_item[r] = d2 * r / n_rng;
        }
        std::vector<int32_t> trc0((size_t)ntiles * n_rng + 1, 0);      // first chunk of (tile, range)
        {   // the tiles are independent (a tile's entries are the ratings of its users: [uptr[tile_u[t]], uptr[tile_u[t + 1]])):
            // built by the host threads side by side, their chunk lists concatenated in tile order afterwards
            std::vector<std::vector<int32_t>> tile_chunks((size_t)ntiles);
            std::vector<std::vector<int32_t>> tile_rc((size_t)ntiles);                 // first chunk of every range inside the tile's list
            pcr_parallel_ranges(ntiles, (int)std::min<int64_t>(pcr_host_threads(), ntiles), [&](int, int64_t t0, int64_t t1) {
                std::vector<int64_t> cur(d2), cut(n_rng + 1);
                for (int64_t t = t0; t < t1; ++t) {
                    std::fill(cur.begin(), cur.end(), 0);
                    for (int64_t z = uptr[tile_u[t]]; z < uptr[tile_u[t + 1]]; ++z) cur[item[z]]++;
                    int64_t run = uptr[tile_u[t]];
                    for (int64_t j = 0; j < d2; ++j) { const int64_t n = cur[j]; cur[j] = run; run += n; }
                    for (int r = 0; r < n_rng; ++r) cut[r] = rng_item[r] < d2 ? cur[rng_item[r]] : run;      // where the tile's entries cross into each item range
                    cut[n_rng] = run;
                    for (int64_t u = tile_u[t]; u < tile_u[t + 1]; ++u)
                        for (int64_t z = uptr[u]; z < uptr[u + 1]; ++z) {
                            const int64_t p = cur[item[z]]++;
                            cpos[z] = (int32_t)p; cuser[p] = (int32_t)u; crow[p] = item[z];
                        }
                    tile_rc[t].resize(n_rng);
                    for (int r = 0; r < n_rng; ++r) {
                        tile_rc[t][r] = (int32_t)tile_chunks[t].size();
                        for (int64_t a = cut[r]; a < cut[r + 1]; a += spmm_chunk) tile_chunks[t].push_back((int32_t)a);   // chunks never straddle tiles or ranges
                    }
                }
            });
            for (int64_t t = 0; t < ntiles; ++t) {
                for (int r = 0; r < n_rng; ++r) trc0[(size_t)t * n_rng + r] = (int32_t)chunk_ptr.size() + tile_rc[t][r];
                chunk_ptr.insert(chunk_ptr.end(), tile_chunks[t].begin(), tile_chunks[t].end());
            }
        }
        trc0[(size_t)ntiles * n_rng] = (int32_t)chunk_ptr.size();
        const int64_t nchunks = (int64_t)chunk_ptr.size();
        chunk_ptr.push_back((int32_t)nnz_local);
        // (chunk, item) incidences in chunk order -> item-major slab rows: the slots of one item are consecutive
        std::vector<int32_t> inc_item;
        inc_base.assign(nchunks + 1, 0); item_slot.assign(d2 + 1, 0);
        cuf = cuser;      // k_spmm's per-entry word: the user id, and in the sign bit "a new item starts here" (never at a chunk start)
        const int nth = pcr_host_threads();
        pcr_parallel_ranges(nchunks, nth, [&](int, int64_t c0, int64_t c1) {          // incidences per chunk, and the new-item flags
            for (int64_t c = c0; c < c1; ++c) {
                int32_t n = 1;
                for (int64_t z = (int64_t)chunk_ptr[c] + 1; z < chunk_ptr[c + 1]; ++z)
                    if (crow[z] != crow[z - 1]) { ++n; cuf[z] |= (int32_t)0x80000000; }
                inc_base[c + 1] = chunk_ptr[c + 1] > chunk_ptr[c] ? n : 0;
            }
        });
        for (int64_t c = 0; c < nchunks; ++c) inc_base[c + 1] += inc_base[c];
        inc_item.resize((size_t)inc_base[nchunks]);
        pcr_parallel_ranges(nchunks, nth, [&](int, int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; ++c) {
                int32_t o = inc_base[c];
                for (int64_t z = chunk_ptr[c]; z < chunk_ptr[c + 1]; ++z)
                    if (z == chunk_ptr[c] || crow[z] != crow[z - 1]) inc_item[o++] = crow[z];
            }
        });
        for (int32_t j : inc_item) item_slot[j + 1]++;
        for (int64_t j = 0; j < d2; ++j) item_slot[j + 1] += item_slot[j];
        slot_id.resize(inc_item.size());
        {
            std::vector<int32_t> nxt(item_slot.begin(), item_slot.end() - 1);
            for (size_t i = 0; i < inc_item.size(); ++i) slot_id[i] = nxt[inc_item[i]]++;
        }
        // workgroup -> chunks: gpb chunks per workgroup, never across tiles; tile t is walked by workgroups b = t mod 8 (mod 8).
        // One plan per item range, back to back (each a multiple of 8 workgroups, so the affinity holds in a launch of one range
        // as in a launch of all of them).
        const int gpb = 256 / geo.G;
        rng_blk.assign(n_rng + 1, 0);
        for (int r = 0; r < n_rng; ++r) {
            std::vector<std::vector<int2>> per_xcd(8);
            size_t rr8 = 0;
            for (int64_t t = 0; t < ntiles; ++t) {
                const int32_t c0 = trc0[(size_t)t * n_rng + r], c1 = trc0[(size_t)t * n_rng + r + 1];
                // tile t -> XCD t mod 8; 2 or 4 tiles: tile t -> the XCDs {t, t + ntiles, ..} in turn (each of them then caches only
                // that tile's rows of U and slice of c); any other count below 8: no affinity, use every XCD
                size_t turn = 0;
                for (int32_t c = c0; c < c1; c += gpb) {
                    const size_t x = ntiles >= 8 ? (size_t)(t % 8) : (ntiles == 2 || ntiles == 4) ? (size_t)t + (size_t)ntiles * (turn++ % (8 / ntiles)) : rr8++ % 8;
                    per_xcd[x].push_back(make_int2(c, std::min<int32_t>(gpb, c1 - c)));
                }
            }
            size_t deepest = 0;
            for (auto& v : per_xcd) deepest = std::max(deepest, v.size());
            const size_t base = blk.size();
            blk.resize(base + deepest * 8, make_int2(0, 0));
            for (int x = 0; x < 8; ++x)
                for (size_t i = 0; i < per_xcd[x].size(); ++i) blk[base + i * 8 + x] = per_xcd[x][i];
            rng_blk[r + 1] = (int)blk.size();
        }
        if (blk.empty()) blk.push_back(make_int2(0, 0));
        P.blocks = rng_blk[n_rng];
        P.slab_rows = inc_item.size();
        P.ntiles = (int)ntiles;
    }
}
#endif
