// dense_kernels.hip -- dense-graph translation unit of the data-parallel tree-search step on gfx950: tree_core.inc
// instantiated with the DenseSpace policy (space_dense.inc), launch-per-phase kernels only (the space's state vector,
// 3E + 1 floats, does not fit the LDS plans of the CU-resident step forms; its evaluator is the batched GEMM).
// Built with -ffp-contract=off like the other tree units.
#include <hip/hip_runtime.h>

#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_dense.inc"

using DSP = DenseSpace<2>;

void dense_launch_init_roots(const Arenas &a, const uint8_t *d_adj, const uint64_t *d_packed, void *stream) {
    k_init_roots<DSP><<<dim3(a.B), dim3(64), DSP::dyn_bytes(a), (hipStream_t)stream>>>(a, d_adj, d_packed);
}
void dense_launch_add_actions(const Arenas &a, int root_mode, void *stream) {
    k_add_actions<DSP><<<dim3(a.B), dim3(64), DSP::dyn_bytes(a), (hipStream_t)stream>>>(a, root_mode);
}
void dense_launch_rollout(const Arenas &a, const TolTable &tol, void *stream) {
    k_rollout<DSP><<<dim3(a.B), dim3(64), DSP::dyn_bytes(a), (hipStream_t)stream>>>(a, tol);
}
void dense_launch_argmin(const Arenas &a, int init_mode, void *stream) {
    k_argmin<DSP><<<dim3(1), dim3(1024), DSP::dyn_bytes(a), (hipStream_t)stream>>>(a, init_mode);
}
void dense_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) {
    k_observe<DSP><<<dim3(a.B), dim3(64), DSP::dyn_bytes(a), (hipStream_t)stream>>>(a, n_obs_tol);
}

} // namespace azd
