// sanitize/device_absent.cpp -- SANITIZER BUILDS ONLY (make -C primalcr_amd/csrc asan): never part of libprimalcr.so.
//
// AddressSanitizer / UBSan run on the CPU build only (GPU sanitizers are not available on this pool), so the sanitized
// library is pcr_host.cpp (loader, parser, cache, partitioner, model file, knob table) + the CLIs' host paths built with g++
// -fsanitize=address,undefined and NO HIP object.  The [device] entry points of include/primalcr.h must still resolve; here
// every one of them does what the real library does on a box without a GPU: set the error message and return
// PCR_ERR_DEVICE.  Nothing is computed -- this is not a CPU path.
#include <cstring>
#include <string>

#include "pcr_host.h"

struct pcr_solver { int unused; };

static int absent() {
    pcr_set_error("no HIP device available: libprimalcr has no CPU fallback for the training path (sanitizer build: host code only)");
    return PCR_ERR_DEVICE;
}

extern "C" {
int pcr_solver_create(const pcr_dataset* ds, const pcr_params* p, int rank, int nranks, pcr_solver** out) {
    if (!ds || !p || !out || nranks < 1 || rank < 0 || rank >= nranks) { pcr_set_error("pcr_solver_create: bad argument"); return PCR_ERR_ARG; }
    return absent();
}
int pcr_solver_create_shard(const pcr_dataset* ds, const pcr_params* p, int rank, int nranks, int64_t first_user, int64_t, pcr_solver** out) {
    if (!ds || !p || !out || nranks < 1 || rank < 0 || rank >= nranks || first_user < 0) { pcr_set_error("pcr_solver_create_shard: bad argument"); return PCR_ERR_ARG; }
    return absent();
}
void pcr_solver_destroy(pcr_solver*) {}
int pcr_device_warmup(int) { return absent(); }
int pcr_comm_unique_id(void* id128) { if (!id128) { pcr_set_error("null id"); return PCR_ERR_ARG; } return absent(); }
#define NO_SOLVER(name, ...) int name(__VA_ARGS__) { pcr_set_error("null solver"); return PCR_ERR_ARG; }
NO_SOLVER(pcr_solver_comm_init, pcr_solver*, const void*)
NO_SOLVER(pcr_solver_comm_init_p2p, pcr_solver*, const char*)
int pcr_solver_comm_nranks(pcr_solver*) { return -1; }
NO_SOLVER(pcr_solver_counter, pcr_solver*, const char*, double*)
NO_SOLVER(pcr_solver_ustep_classes, pcr_solver*, char*, int64_t)
NO_SOLVER(pcr_solver_setup_phase, const pcr_solver*, int, const char**, double*)
NO_SOLVER(pcr_solver_set_local_only, pcr_solver*, int)
NO_SOLVER(pcr_solver_shard, const pcr_solver*, int64_t*, int64_t*, int64_t*)
NO_SOLVER(pcr_solver_set_factors, pcr_solver*, const double*, const double*)
NO_SOLVER(pcr_solver_get_factors, pcr_solver*, double*, double*)
NO_SOLVER(pcr_solver_set_factors_local, pcr_solver*, const double*, const double*)
NO_SOLVER(pcr_solver_get_factors_local, pcr_solver*, double*, double*)
NO_SOLVER(pcr_comp_m, pcr_solver*, double*)
NO_SOLVER(pcr_objective, pcr_solver*, double*)
NO_SOLVER(pcr_obtain_g, pcr_solver*, double*)
NO_SOLVER(pcr_compute_Ha, pcr_solver*, const double*, double*)
NO_SOLVER(pcr_solve_delta, pcr_solver*, const double*, double*, int*)
NO_SOLVER(pcr_update_V, pcr_solver*, double*, int*)
NO_SOLVER(pcr_update_U, pcr_solver*, double*, int64_t*)
NO_SOLVER(pcr_evaluate, pcr_solver*, int, int, double*, double*)
NO_SOLVER(pcr_train, pcr_solver*, pcr_log_fn, void*, pcr_iter_stats*)
NO_SOLVER(pcr_iterate, pcr_solver*, int, pcr_iter_stats*)
NO_SOLVER(pcr_solver_sync, pcr_solver*)
NO_SOLVER(pcr_profile_enable, pcr_solver*, int)
NO_SOLVER(pcr_profile_list, pcr_solver*, char*, int64_t)
NO_SOLVER(pcr_profile_get, pcr_solver*, const char*, double*, int64_t*)
NO_SOLVER(pcr_profile_launches, pcr_solver*, const char*, int64_t*)
NO_SOLVER(pcr_profile_scope, pcr_solver*, const char*, int64_t*, int64_t*)
NO_SOLVER(pcr_profile_reset, pcr_solver*)
int pcr_predict(const double* U, int64_t, const double* V, int64_t, int64_t k, int64_t n, const int32_t* user, const int32_t* item, double* pred, int) {
    if (!U || !V || k < 1 || n < 0 || (n > 0 && (!user || !item || !pred))) { pcr_set_error("pcr_predict: bad argument"); return PCR_ERR_ARG; }
    return absent();
}
}
