// pcr_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the PrimalCR /
// PrimalCR++ hot path.  Header-only templates, instantiated in pcr_solver.hip:
//   pcr_prims.h (wave / team primitives, sorts, sweep coefficients, row primitives), pcr_vside.h (V step), pcr_ustep.h (U step),
//   pcr_eval.h (evaluator, objective reductions, predict); pcr_gram.h (optional dual-form U step on MFMA).
//
// Kernel map (reference site -> kernel); design and rooflines: DESIGN.md section 3:
//   K1        comp_m_new, b = u.a in compute_Ha_new        -> k_sddmm     (rating-parallel)
//   K2+K6     get_sorted_mm + objective_new                -> k_prepare   (per user: sort, windows, loss)
//   K4        obtain_g_new  sweep (pcrpp.cpp:214-238)      -> k_vsweep<GRAD> / k_vsweep_wave + k_spmm + k_spmm_fin
//   K5        compute_Ha_new sweep (pcrpp.cpp:294-318)     -> k_vsweep<HV>   / k_vsweep_wave + k_spmm + k_spmm_fin
//   K7        solve_delta_new vector ops (pcrpp.cpp:335)   -> k_cg_init / k_cg_bc (k_cg_a with a communicator)
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
#include "pcr_prims.h"
#include "pcr_vside.h"
#include "pcr_ustep.h"
#include "pcr_eval.h"
