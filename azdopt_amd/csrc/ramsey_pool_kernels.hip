// ramsey_pool_kernels.hip -- the pool step (pool_step.inc) for the Ramsey space, in its own translation unit.
#define AZD_TU_ASYNC 1
#define AZD_TU_POOL 1
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_ramsey.inc"
#include "persistent_step.inc"
#include "async_step.inc"
#include "pool_step.inc"

#define DISPATCH_RKW(A, FN, ...)                                  \
    switch ((A).KW) {                                             \
    case 1: FN<RamseySpace<1>>(__VA_ARGS__); break;               \
    case 2: FN<RamseySpace<2>>(__VA_ARGS__); break;               \
    case 3: FN<RamseySpace<3>>(__VA_ARGS__); break;               \
    case 4: FN<RamseySpace<4>>(__VA_ARGS__); break;               \
    case 5: FN<RamseySpace<5>>(__VA_ARGS__); break;               \
    default: FN<RamseySpace<6>>(__VA_ARGS__); break;              \
    }

template <class SP>
static void l_pool(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key, const float *params,
                   const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    if (hipFuncSetAttribute((const void *)k_pool<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return;
    (void)hipMemsetAsync(log_key, 0xFF, (size_t)n_calls * sizeof(unsigned long long), st);
    k_pool<SP><<<dim3(n_blocks), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, n_calls, log_key, dyn_stride, params, a.state_vecs, a.h_theta, wpk);
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, n_calls, log_key);
}
void ramsey_launch_pool(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key, const float *params,
                        const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_RKW(a, l_pool, a, d_args, n_calls, log_key, params, wpk, n_blocks, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
bool ramsey_pool_plan(const Arenas &a, const FusedEval &ev, PoolArgs *pool, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    return pool_plan_common(a, ev, pool, dyn_stride, dyn_bytes, why, RamseySpace<1>::pool_dyn_bytes(a), sizeof(RamseyLds));
}

} // namespace azd
