// dense_kernels.hip -- dense-graph translation unit of the data-parallel tree-search step on gfx950: tree_core.inc
// instantiated with the DenseSpace policy (space_dense.inc) for key widths 2 / 4 / 10 / 16, launch-per-phase kernels and the
// device root policy (the space's state vector,
// 3E + 1 floats, does not fit the LDS plans of the CU-resident step forms; its evaluator is the batched GEMM).
// Built with -ffp-contract=off like the other tree units.
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_dense.inc"

#include "root_policy.inc"

// the key width (words of a rank set) follows the engine's max_slots: 2, 4, 10 or 16 (engine.hip)
#define DISPATCH_DKW(A, FN, ...)                                  \
    switch ((A).KW) {                                             \
    case 2: FN<DenseSpace<2>>(__VA_ARGS__); break;                \
    case 4: FN<DenseSpace<4>>(__VA_ARGS__); break;                \
    case 10: FN<DenseSpace<10>>(__VA_ARGS__); break;              \
    default: FN<DenseSpace<16>>(__VA_ARGS__); break;              \
    }

template <class SP>
static void l_init_roots(const Arenas &a, const uint8_t *p, const uint64_t *m, hipStream_t st) {
    k_init_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, p, m);
}
template <class SP>
static void l_add_actions(const Arenas &a, int root_mode, hipStream_t st) {
    k_add_actions<SP><<<dim3(a.tn ? a.tn : a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, root_mode);
}
template <class SP>
static void l_rollout(const Arenas &a, const TolTable &tol, hipStream_t st) {
    k_rollout<SP><<<dim3(a.tn ? a.tn : a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, tol);
}
template <class SP>
static void l_argmin(const Arenas &a, int init_mode, hipStream_t st) {
    k_argmin<SP><<<dim3(1), dim3(1024), SP::dyn_bytes(a), st>>>(a, init_mode);
}
template <class SP>
static void l_observe(const Arenas &a, uint32_t tol, hipStream_t st) {
    k_observe<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, tol);
}
template <class SP>
static void l_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax, uint8_t *d_adj,
                           uint64_t *d_packed, uint64_t *d_slots, hipStream_t st) {
    k_modify_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, seed, epoch, first_agent, kmin, kmax, d_adj, d_packed, d_slots);
}

void dense_launch_init_roots(const Arenas &a, const uint8_t *d_adj, const uint64_t *d_packed, void *stream) {
    DISPATCH_DKW(a, l_init_roots, a, d_adj, d_packed, (hipStream_t)stream);
}
void dense_launch_add_actions(const Arenas &a, int root_mode, void *stream) { DISPATCH_DKW(a, l_add_actions, a, root_mode, (hipStream_t)stream); }
void dense_launch_rollout(const Arenas &a, const TolTable &tol, void *stream) { DISPATCH_DKW(a, l_rollout, a, tol, (hipStream_t)stream); }
void dense_launch_argmin(const Arenas &a, int init_mode, void *stream) { DISPATCH_DKW(a, l_argmin, a, init_mode, (hipStream_t)stream); }
template <class SP>
static void l_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, hipStream_t st) {
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, n_calls, log_key, nullptr);
}
void dense_launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream) {
    DISPATCH_DKW(a, l_argmin_log, a, n_calls, log_key, (hipStream_t)stream);
}
void dense_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) { DISPATCH_DKW(a, l_observe, a, n_obs_tol, (hipStream_t)stream); }
void dense_launch_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                               uint8_t *d_adj, uint64_t *d_packed, uint64_t *d_slots, void *stream) {
    DISPATCH_DKW(a, l_modify_roots, a, seed, epoch, first_agent, kmin, kmax, d_adj, d_packed, d_slots, (hipStream_t)stream);
}

} // namespace azd
